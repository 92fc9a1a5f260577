// pmf_comm_fit.hip -- the cross-rank exchange of the sharded fit (pmf_comm_*) and the epoch loop (pmf_fit).
#include <rccl/rccl.h>   // types and enums only: the entry points are resolved with dlopen at pmf_comm_init (no link dependency)
#include <dlfcn.h>

#include "pmf_ctx.h"

// ------------------------------------------------------------------------------------------------
// cross-rank exchange (no reference counterpart: the reference is single-GPU, one process per GPU being its
// habit for independent fits, analyses/scripts/julia/script_util.jl:278-306)
// ------------------------------------------------------------------------------------------------
struct RcclApi {
  void *dl = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommInitRankConfig)(ncclComm_t *, int, ncclUniqueId, int, ncclConfig_t *) = nullptr;   // optional
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  const char *(*GetLastError)(ncclComm_t) = nullptr;
};
static RcclApi g_rccl;
// RCCL is loaded on first use (dlopen): a single-GPU host never maps the 570 MB library, and libpmf_hip.so has no
// link-time dependency on it.  librccl.so.1 resolves to the ROCm installation the library's own HIP runtime comes from.
static int rccl_load() {
  if (g_rccl.dl) return 0;
  // The RCCL that belongs to THIS library's HIP runtime: the librccl.so.1 next to the libamdhip64 our HIP calls are bound
  // to.  (A process that also holds PyTorch-ROCm has a second HIP runtime and a second RCCL, torch's bundled ones; a
  // stream of one runtime must not reach the other's RCCL.)  PMF_RCCL_LIB overrides; plain names are the fallback.
  std::string sibling;
  {
    Dl_info info;
    if (dladdr((void *)&hipStreamCreateWithFlags, &info) && info.dli_fname) {
      std::string p(info.dli_fname);
      const size_t k = p.rfind('/');
      if (k != std::string::npos) sibling = p.substr(0, k + 1) + "librccl.so.1";
    }
  }
  const char *names[] = {getenv("PMF_RCCL_LIB"), sibling.empty() ? nullptr : sibling.c_str(), "librccl.so.1", "librccl.so"};
  void *dl = nullptr;
  for (const char *nm : names) {
    if (!nm || !*nm) continue;
    dl = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
    if (dl) break;
  }
  if (!dl) return pmf_fail("cannot load librccl.so.1 (%s): multi-GPU fits need RCCL", dlerror());
  RcclApi r;
  r.dl = dl;
#define PMF_SYM(field, name)                                                          \
  *(void **)(&r.field) = dlsym(dl, name);                                             \
  if (!r.field) return pmf_fail("librccl: symbol %s not found", name)
  PMF_SYM(GetUniqueId, "ncclGetUniqueId");
  PMF_SYM(CommInitRank, "ncclCommInitRank");
  PMF_SYM(CommDestroy, "ncclCommDestroy");
  PMF_SYM(AllReduce, "ncclAllReduce");
  PMF_SYM(GroupStart, "ncclGroupStart");
  PMF_SYM(GroupEnd, "ncclGroupEnd");
  PMF_SYM(GetErrorString, "ncclGetErrorString");
#undef PMF_SYM
  *(void **)(&r.GetLastError) = dlsym(dl, "ncclGetLastError");
  *(void **)(&r.CommInitRankConfig) = dlsym(dl, "ncclCommInitRankConfig");
  g_rccl = r;
  return 0;
}
static int rccl_chk(ncclResult_t rc, const char *what, ncclComm_t comm = nullptr) {
  if (rc == ncclSuccess) return 0;
  const char *detail = (g_rccl.GetLastError && comm) ? g_rccl.GetLastError(comm) : "";
  return pmf_fail("%s failed: %s %s", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "?", detail ? detail : "");
}

extern "C" int pmf_comm_get_unique_id(void *id_out) {
  if (!id_out) return pmf_fail("null id buffer");
  PMFCHK(rccl_load());
  ncclUniqueId id;
  PMFCHK(rccl_chk(g_rccl.GetUniqueId(&id), "ncclGetUniqueId"));
  static_assert(sizeof(ncclUniqueId) == PMF_COMM_ID_BYTES, "ncclUniqueId size");
  memcpy(id_out, &id, sizeof(id));
  return 0;
}

bool comm_active(const pmf_ctx *c) { return c->comm.nccl != nullptr || c->comm.host_fn != nullptr; }

int comm_release(pmf_ctx *c) {
  Comm &m = c->comm;
  if (m.stream) (void)hipStreamSynchronize(m.stream);
  if (m.nccl && g_rccl.CommDestroy) (void)g_rccl.CommDestroy((ncclComm_t)m.nccl);
  m.nccl = nullptr;
  m.host_fn = nullptr;
  m.host_user = nullptr;
  for (auto e : m.ev_ready) (void)hipEventDestroy(e);
  for (auto e : m.ev_done) (void)hipEventDestroy(e);
  m.ev_ready.clear(); m.ev_done.clear();
  hipEvent_t *evs[4] = {&m.ev_loss_ready, &m.ev_loss_done, &m.ev_layer_ready, &m.ev_layer_done};
  for (auto pe : evs) { if (*pe) (void)hipEventDestroy(*pe); *pe = nullptr; }
  if (m.stream) (void)hipStreamDestroy(m.stream);
  m.stream = nullptr;
  if (m.stage) (void)hipHostFree(m.stage);
  m.stage = nullptr; m.stage_bytes = 0;
  if (m.dstage) (void)hipFree(m.dstage);
  m.dstage = nullptr; m.dstage_bytes = 0;
  m.rank = 0; m.nranks = 1; m.reserve_cus = 0; m.cta_cap = 0; m.broken = false; m.m_mean = 0;
  return 0;
}
static int comm_common_init(pmf_ctx *c, int rank, int nranks) {
  if (nranks < 1 || rank < 0 || rank >= nranks) return pmf_fail("bad rank %d of %d", rank, nranks);
  comm_release(c);
  Comm &m = c->comm;
  m.rank = rank; m.nranks = nranks;
  HIPCHK(hipStreamCreateWithFlags(&m.stream, hipStreamNonBlocking));
  hipEvent_t *evs[4] = {&m.ev_loss_ready, &m.ev_loss_done, &m.ev_layer_ready, &m.ev_layer_done};
  for (auto pe : evs) HIPCHK(hipEventCreateWithFlags(pe, hipEventDisableTiming));
  return 0;
}

extern "C" int pmf_comm_init(pmf_ctx *c, int rank, int nranks, const void *unique_id) {
  PMFCHK(ctx_bind(c));
  if (!unique_id) return pmf_fail("null unique id");
  PMFCHK(rccl_load());
  // The fused pass is a persistent grid that fills every CU: a collective's kernels get a CU only when a workgroup
  // retires.  With more than one rank the pass therefore leaves PMF_COMM_CTAS CUs (default 4) free, and THIS communicator
  // is held to as many workgroups through its own configuration (ncclConfig_t.maxCTAs, ncclCommInitRankConfig): nothing
  // process-wide is touched -- no environment variable is set, other communicators of the host keep RCCL's defaults.
  // PMF_COMM_CTAS=0: no reservation and RCCL's own channel count.  A host that has set NCCL_MAX_NCHANNELS itself below
  // the reservation gets the smaller of the two from RCCL; the reservation then merely leaves a few CUs idle.
  int ctas = 4;
  if (const char *e = getenv("PMF_COMM_CTAS")) ctas = atoi(e);
  ctas = std::max(0, std::min(ctas, c->n_cu / 4));
  PMFCHK(comm_common_init(c, rank, nranks));
  ncclUniqueId id;
  memcpy(&id, unique_id, sizeof(id));
  ncclComm_t comm = nullptr;
  int rc;
  bool capped = false;
  const bool cap_one = getenv("PMF_COMM_CAP_ONE_RANK") != nullptr;   // (tests: the configured init on a one-rank communicator)
  if ((nranks > 1 || cap_one) && ctas > 0 && g_rccl.CommInitRankConfig) {
    ncclConfig_t cfg = NCCL_CONFIG_INITIALIZER;
    cfg.maxCTAs = ctas;   // (minCTAs stays undefined = RCCL's default of 1)
    rc = rccl_chk(g_rccl.CommInitRankConfig(&comm, nranks, id, rank, &cfg), "ncclCommInitRankConfig");
    capped = true;
  } else {
    rc = rccl_chk(g_rccl.CommInitRank(&comm, nranks, id, rank), "ncclCommInitRank");
  }
  if (rc < 0) { comm_release(c); return rc; }
  c->comm.nccl = comm;
  // without the cap (an RCCL too old for ncclCommInitRankConfig) the CUs are still left free: the collective's first
  // workgroups start beside the pass, the rest when workgroups of the pass retire -- correct, less overlap
  c->comm.reserve_cus = nranks > 1 ? ctas : 0;
  c->comm.cta_cap = capped ? ctas : 0;
  return 0;
}

extern "C" int pmf_comm_init_host(pmf_ctx *c, int rank, int nranks, pmf_host_allreduce_fn fn, void *user) {
  PMFCHK(ctx_bind(c));
  if (!fn) return pmf_fail("null all-reduce callback");
  PMFCHK(comm_common_init(c, rank, nranks));
  c->comm.host_fn = fn;
  c->comm.host_user = user;
  return 0;
}

extern "C" int pmf_comm_destroy(pmf_ctx *c) {
  PMFCHK(ctx_bind(c));
  return comm_release(c);
}

extern "C" int pmf_comm_set_chunks(pmf_ctx *c, int n_chunks) {
  if (!c) return pmf_fail("null context");
  if (n_chunks < 0 || n_chunks > PMF_MAX_CHUNKS) return pmf_fail("n_chunks=%d out of range (0..%d)", n_chunks, PMF_MAX_CHUNKS);
  c->n_chunks_req = n_chunks;
  return 0;
}

extern "C" int pmf_comm_info(pmf_ctx *c, int *rank, int *nranks, int *transport, int *n_chunks, int *reserved_cus,
                             int64_t *n_collectives) {
  if (!c) return pmf_fail("null context");
  if (rank) *rank = c->comm.rank;
  if (nranks) *nranks = c->comm.nranks;
  if (transport) *transport = c->comm.nccl ? PMF_COMM_RCCL : (c->comm.host_fn ? PMF_COMM_HOST : PMF_COMM_NONE);
  if (n_chunks) *n_chunks = c->last_chunks;
  if (reserved_cus) *reserved_cus = c->comm.nranks > 1 ? c->comm.reserve_cus : 0;
  if (n_collectives) *n_collectives = c->comm.n_allreduce;
  return 0;
}

// in-place sum over the ranks of `count` elements at device address p, ordered on the communication stream
static int comm_allreduce(pmf_ctx *c, void *p, int64_t count, bool f64) {
  Comm &m = c->comm;
  if (count <= 0) return 0;
  m.n_allreduce++;
  if (m.nccl)
    return rccl_chk(g_rccl.AllReduce(p, p, (size_t)count, f64 ? ncclFloat64 : ncclFloat32, ncclSum, (ncclComm_t)m.nccl, m.stream),
                    "ncclAllReduce", (ncclComm_t)m.nccl);
  // host-staged transport (tests): device -> pinned host -> callback (e.g. gloo) -> device, blocking
  const size_t bytes = (size_t)count * (f64 ? 8 : 4);
  if (bytes > m.stage_bytes) {
    if (m.stage) (void)hipHostFree(m.stage);
    m.stage = nullptr; m.stage_bytes = 0;
    HIPCHK(hipHostMalloc(&m.stage, bytes));
    m.stage_bytes = bytes;
  }
  HIPCHK(hipMemcpyAsync(m.stage, p, bytes, hipMemcpyDeviceToHost, m.stream));
  HIPCHK(hipStreamSynchronize(m.stream));
  if (m.host_fn(m.host_user, m.stage, count, f64 ? 1 : 0) != 0) return pmf_fail("host all-reduce callback failed");
  HIPCHK(hipMemcpyAsync(p, m.stage, bytes, hipMemcpyHostToDevice, m.stream));
  HIPCHK(hipStreamSynchronize(m.stream));
  return 0;
}
// Sum (op 0) or maximum (op 1) over the ranks of a HOST buffer through the communicator, for what the host keeps
// between the GD stages: the column / batch statistics of pmf_stats, timings, flags.  Blocking.
extern "C" int pmf_comm_allreduce(pmf_ctx *c, void *host_buf, int64_t count, int dtype, int op) {
  PMFCHK(ctx_bind(c));
  if (!host_buf || count < 0) return pmf_fail("bad buffer");
  if (dtype != 0 && dtype != 1) return pmf_fail("dtype must be 0 (float32) or 1 (float64)");
  if (op != 0 && op != 1) return pmf_fail("op must be 0 (sum) or 1 (max)");
  if (!comm_active(c) || count == 0) return 0;   // one rank: nothing to do
  Comm &m = c->comm;
  const size_t bytes = (size_t)count * (dtype ? 8 : 4);
  if (m.nccl) {
    if (bytes > m.dstage_bytes) {   // device staging buffer of the communicator, grown on demand (no allocation per call)
      if (m.dstage) (void)hipFree(m.dstage);
      m.dstage = nullptr; m.dstage_bytes = 0;
      const size_t want = std::max<size_t>(bytes, 1 << 16);
      HIPCHK(hipMalloc(&m.dstage, want));
      m.dstage_bytes = want;
    }
    void *d = m.dstage;
    hipError_t e = hipMemcpyAsync(d, host_buf, bytes, hipMemcpyHostToDevice, m.stream);
    int rc = 0;
    if (e == hipSuccess)
      rc = rccl_chk(g_rccl.AllReduce(d, d, (size_t)count, dtype ? ncclFloat64 : ncclFloat32, op ? ncclMax : ncclSum, (ncclComm_t)m.nccl, m.stream),
                    "ncclAllReduce", (ncclComm_t)m.nccl);
    if (e == hipSuccess && rc == 0) e = hipMemcpyAsync(host_buf, d, bytes, hipMemcpyDeviceToHost, m.stream);
    if (e == hipSuccess) e = hipStreamSynchronize(m.stream);
    if (rc < 0) return rc;
    if (e != hipSuccess) return pmf_fail("pmf_comm_allreduce: %s", hipGetErrorString(e));
    m.n_allreduce++;
    return 0;
  }
  if (op != 0) return pmf_fail("the host-staged transport only sums");
  if (m.host_fn(m.host_user, host_buf, count, dtype) != 0) return pmf_fail("host all-reduce callback failed");
  m.n_allreduce++;
  return 0;
}

// the communication stream continues after everything enqueued on the compute stream so far / vice versa
static int comm_after_compute(pmf_ctx *c, hipEvent_t ev) {
  HIPCHK(hipEventRecord(ev, c->stream));
  HIPCHK(hipStreamWaitEvent(c->comm.stream, ev, 0));
  return 0;
}
static int comm_mark(pmf_ctx *c, hipEvent_t ev) {
  HIPCHK(hipEventRecord(ev, c->comm.stream));
  return 0;
}
static int compute_after_comm(pmf_ctx *c, hipEvent_t ev) {
  HIPCHK(hipStreamWaitEvent(c->stream, ev, 0));
  return 0;
}

// MF.fit!(model.matfac, model.data; ...) (src/fit.jl:24-36): the epoch loop.
//
// One epoch = data pass (fused kernel over S column chunks [+ layer pass]) -> X step -> Y step per chunk [-> layer steps]
// -> loss.  The loop is software-pipelined across epochs: the data pass of epoch e+1 is launched BEFORE the host has
// seen the loss of epoch e -- chunk s of epoch e+1 right after the Y step of chunk s of epoch e, which is all it depends
// on besides the X step -- so that
//   * the host's wait for the loss (and its termination test) runs beside the next data pass instead of idling the GPU,
//   * with a communicator, the all-reduce of chunk s's grad(Y) has until the Y step of chunk s to finish, i.e. it runs
//     beside the launches of chunks s+1 .. S-1 of this epoch and 0 .. s-1 of the next: no collective sits on the
//     critical path (one exchange step per epoch: S grad(Y) slices, the local loss as two doubles, and the layer
//     gradients when they train).
// A data pass only writes gradient buffers, never parameters: when epoch e terminates the fit, the speculative pass of
// e+1 is simply dropped and the parameters are exactly those after epoch e's steps (same results as the plain loop).
static int fit_loop(pmf_ctx *c, const pmf_fit_opts *o, pmf_fit_result *res);

// Error exits.  Inside the pipelined loop a speculative data pass and collectives may already be enqueued when a call
// fails.  Every error exit therefore drains both streams before returning (the context's buffers are not written behind
// the caller's back), and with a communicator of more than one rank the communicator is marked unusable: the peers may be
// blocked in a collective this rank never issued, so a failed rank is fatal for the group -- the host must end the
// process (or destroy the communicator on every rank); further pmf_fit calls on it are refused.
extern "C" int pmf_fit(pmf_ctx *c, const pmf_fit_opts *o, pmf_fit_result *res) {
  PMFCHK(ctx_bind(c));
  PMFCHK(check_ready(c));
  if (!o || !res) return pmf_fail("null opts/result");
  if (c->comm.broken) return pmf_fail("the communicator is unusable after a failed fit: pmf_comm_destroy it on every rank");
  const int rc = fit_loop(c, o, res);
  if (rc < 0) {
    const std::string msg = pmf_last_error();
    (void)hipStreamSynchronize(c->stream);
    if (c->comm.stream) (void)hipStreamSynchronize(c->comm.stream);
    (void)hipGetLastError();
    if (comm_active(c) && c->comm.nranks > 1) c->comm.broken = true;
    pmf_fail("%s%s", msg.c_str(), c->comm.broken ? " [rank failure: the communicator is now unusable]" : "");
  }
  return rc;
}

static int fit_loop(pmf_ctx *c, const pmf_fit_opts *o, pmf_fit_result *res) {
  const auto t0 = std::chrono::steady_clock::now();
  const bool ux = o->update_X != 0, uy = o->update_Y != 0, ul = o->update_col_layers != 0;
  const bool fused = ux || uy || !ul;
  const bool cm = comm_active(c);
  // The column-chunk count must be the same on every rank (it is the number and size of the grad(Y) collectives).  The
  // automatic choice is therefore made from rank-invariant numbers only: N, K and the MEAN rows per rank, which costs one
  // tiny all-reduce per pmf_fit (not per epoch); shards of different heights then still agree.
  if (cm && c->comm.nranks > 1 && c->n_chunks_req <= 0 && fused && uy && !ul) {
    double *hl = c->h_loss + 6;   // (pinned; slots 0..4 carry the loss)
    hl[0] = (double)c->M;
    HIPCHK(hipMemcpyAsync(c->d_loss + 6, hl, sizeof(double), hipMemcpyHostToDevice, c->comm.stream));
    PMFCHK(comm_allreduce(c, c->d_loss + 6, 1, true));
    HIPCHK(hipMemcpyAsync(hl, c->d_loss + 6, sizeof(double), hipMemcpyDeviceToHost, c->comm.stream));
    HIPCHK(hipStreamSynchronize(c->comm.stream));
    c->comm.m_mean = (int64_t)std::llround(hl[0] / c->comm.nranks);
  }
  // (the first epoch is opened before the geometry is chosen: prepare() decides whether the dense batch table exists,
  //  which decides the kernel variant and with it the row-panel height)
  if (o->epoch <= o->max_epochs) PMFCHK(epoch_open(c, o));
  FusedGeom g;
  if (fused) g = fused_geometry(c, ux, uy, /*allow_chunks=*/!ul);
  const int S = fused ? g.S : 1;
  c->last_chunks = S;
  Comm &m = c->comm;
  if (cm) {
    while ((int)m.ev_ready.size() < S) {
      hipEvent_t e0, e1;
      HIPCHK(hipEventCreateWithFlags(&e0, hipEventDisableTiming));
      HIPCHK(hipEventCreateWithFlags(&e1, hipEventDisableTiming));
      m.ev_ready.push_back(e0);
      m.ev_done.push_back(e1);
    }
  }
  if (!c->ev_host) HIPCHK(hipEventCreateWithFlags(&c->ev_host, hipEventDisableTiming));
  int term = PMF_TERM_MAX_EPOCHS, tol_iters = 0, n = 0, last_epoch = o->epoch - 1;
  double prev = 0.0, loss = 0.0;
  const int tol_max = o->tol_max_iters > 0 ? o->tol_max_iters : 3;

  // chunk s of an epoch's data pass, with the exchange of what it completes
  auto pass_chunk = [&](int s) -> int {
    if (fused) {
      PMFCHK(launch_fused_chunk(c, g, s, ux, uy));
      if (cm && uy) {
        const int64_t col0 = g.ct0[s] * 32, col1 = std::min<int64_t>(c->N, (g.ct0[s] + g.nct[s]) * 32);
        PMFCHK(comm_after_compute(c, m.ev_ready[(size_t)s]));
        PMFCHK(comm_allreduce(c, c->P[1].g + col0 * c->Kp, (col1 - col0) * c->Kp, false));
        PMFCHK(comm_mark(c, m.ev_done[(size_t)s]));
      }
    }
    if (ul && s == S - 1) {
      PMFCHK(epoch_layer_pass(c, o, !fused));
      if (cm) {
        PMFCHK(comm_after_compute(c, m.ev_layer_ready));
        if (m.nccl) PMFCHK(rccl_chk(g_rccl.GroupStart(), "ncclGroupStart"));
        for (int w = 2; w < 6; ++w) PMFCHK(comm_allreduce(c, c->P[w].g, c->P[w].n, false));
        if (m.nccl) PMFCHK(rccl_chk(g_rccl.GroupEnd(), "ncclGroupEnd", (ncclComm_t)m.nccl));
        PMFCHK(comm_mark(c, m.ev_layer_done));
      }
    }
    return 0;
  };

  bool in_flight = false;   // a data pass whose epoch has not been finished is enqueued
  if (o->epoch <= o->max_epochs) {
    if (fused) PMFCHK(prepare_fused_pass(c, g, ux, uy));
    for (int s = 0; s < S; ++s) PMFCHK(pass_chunk(s));
    in_flight = true;
  }
  for (int epoch = o->epoch; epoch <= o->max_epochs; ++epoch) {
    const bool more = epoch < o->max_epochs;
    RegCounts rc;
    for (int q = 0; q < 4; ++q) rc.c[q] = 0;
    // ---- X step (row-local) and the rank-local part of the loss: data term + X regularizer
    if (ux) PMFCHK(step_param(c, 0, true, true, 0, &rc.c[0]));
    PMFCHK(launch_loss_reduce(c, rc, 0x03));
    if (cm) {
      PMFCHK(comm_after_compute(c, m.ev_loss_ready));
      PMFCHK(comm_allreduce(c, c->d_loss, 2, true));
      PMFCHK(comm_mark(c, m.ev_loss_done));
    }
    in_flight = false;
    // ---- replicated parameters, chunk by chunk; the next epoch's chunk follows its Y step
    for (int s = 0; s < S; ++s) {
      if (uy) {
        if (cm) PMFCHK(compute_after_comm(c, m.ev_done[(size_t)s]));
        const int64_t col0 = fused ? g.ct0[s] * 32 : 0;
        const int64_t col1 = fused ? std::min<int64_t>(c->N, (g.ct0[s] + g.nct[s]) * 32) : c->N;
        PMFCHK(step_param_range(c, 1, col0 * c->Kp, (col1 - col0) * c->Kp, true, true, 1, &rc.c[1], REG_SLOTS / S, s == S - 1));
      }
      if (s == S - 1) {
        if (ul) {
          if (cm) PMFCHK(compute_after_comm(c, m.ev_layer_done));
          PMFCHK(step_layers(c, o, &rc.c[2]));
        }
        if (cm) PMFCHK(compute_after_comm(c, m.ev_loss_done));
        PMFCHK(launch_loss_reduce(c, rc, 0x1c));
        HIPCHK(hipMemcpyAsync(c->h_loss, c->d_loss, sizeof(double) * 5, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipEventRecord(c->ev_host, c->stream));
      }
      if (more) {
        if (s == 0 && S > 1) PMFCHK(epoch_open(c, o));
        if (s == S - 1 && S == 1) PMFCHK(epoch_open(c, o));
        PMFCHK(pass_chunk(s));
        in_flight = true;
      }
    }
    HIPCHK(hipEventSynchronize(c->ev_host));
    loss = c->h_loss[0] + c->h_loss[1] + (c->h_loss[2] + c->h_loss[3]);
    if (getenv("PMF_DEBUG_LOSS"))
      fprintf(stderr, "[pmf rank %d] epoch %d: data %.10g xreg %.10g yreg %.10g layers %.10g (n_macro %lld, reg counts %d %d %d)\n", m.rank, epoch,
              c->h_loss[0], c->h_loss[1], c->h_loss[2], c->h_loss[3], (long long)c->n_macro, rc.c[0], rc.c[1], rc.c[2]);
    if (getenv("PMF_DEBUG_LOSS") && !std::isfinite(c->h_loss[2])) {
      HIPCHK(hipDeviceSynchronize());
      auto dump = [&](const char *nm, const float *d, int64_t cnt) {
        std::vector<float> h((size_t)cnt);
        (void)hipMemcpy(h.data(), d, sizeof(float) * (size_t)cnt, hipMemcpyDeviceToHost);
        double mn = 1e300, mx = -1e300; int64_t bad = 0, first = -1;
        for (int64_t e = 0; e < cnt; ++e) { if (!std::isfinite(h[(size_t)e])) { if (first < 0) first = e; ++bad; } else { mn = std::min<double>(mn, h[(size_t)e]); mx = std::max<double>(mx, h[(size_t)e]); } }
        fprintf(stderr, "[pmf rank %d]   %s: n %lld min %g max %g nonfinite %lld (first at %lld)\n", m.rank, nm, (long long)cnt, mn, mx, (long long)bad, (long long)first);
      };
      dump("Y", c->P[1].p, c->P[1].n); dump("gY", c->P[1].g, c->P[1].n); dump("accY", c->P[1].acc, c->P[1].n);
      if (c->ard_beta) { dump("beta", c->ard_beta, c->P[1].n); dump("alpha", c->ard_alpha, c->N); }
      std::vector<double> rp((size_t)rc.c[1]);
      (void)hipMemcpy(rp.data(), c->reg_partial + REG_SLOTS, sizeof(double) * rp.size(), hipMemcpyDeviceToHost);
      for (size_t q = 0; q < rp.size(); ++q) if (!std::isfinite(rp[q])) fprintf(stderr, "[pmf rank %d]   yreg partial %zu = %g\n", m.rank, q, rp[q]);
    }
    if (res->loss_trace && n < res->trace_cap) res->loss_trace[n] = loss;
    ++n;
    last_epoch = epoch;
    if (o->verbosity > 0 && o->print_iter > 0 && (epoch % o->print_iter == 0))
      fprintf(stderr, "(%d) Loss=%.8g\n", epoch, loss);
    if (!std::isfinite(loss)) { term = PMF_TERM_NONFINITE; break; }
    if (n > 1) {
      const double diff = prev - loss;
      if (diff < 0) { term = PMF_TERM_LOSS_INCREASE; break; }
      int which = -1;
      if (std::fabs(diff) < o->abs_tol) which = PMF_TERM_ABS_TOL;
      else if (std::fabs(diff / loss) < o->rel_tol) which = PMF_TERM_REL_TOL;
      if (which >= 0) {
        if (++tol_iters >= tol_max) { term = which; break; }
      } else {
        tol_iters = 0;
      }
    }
    prev = loss;
  }
  // a speculative data pass (and its collectives: every rank took the same decision on the same loss, so every rank
  // enqueued them) may still be running: the call returns with both streams idle
  (void)in_flight;
  HIPCHK(hipStreamSynchronize(c->stream));
  if (cm) HIPCHK(hipStreamSynchronize(m.stream));
  harvest_events(c);
  res->term_code = term;
  res->epochs = last_epoch;
  res->n_trace = res->loss_trace ? std::min(n, res->trace_cap) : 0;
  res->final_loss = loss;
  res->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  return 0;
}
