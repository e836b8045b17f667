// gvec_api.hip — the C ABI of include/generals_vec.h over the HIP kernels.
// Plain HIP runtime only (no torch, no CPU fallback): every compute entry point launches
// gfx950 kernels and fails with GVEC_E_NO_DEVICE when there is no GPU.
#include <hip/hip_runtime.h>

#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "gvec_launch.hpp"

using namespace gvec;

static thread_local char g_err[512] = "";
static void set_err(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
}

#define HIPCHK(expr)                                                                   \
  do {                                                                                 \
    hipError_t e__ = (expr);                                                           \
    if (e__ != hipSuccess) {                                                           \
      set_err("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
      (void)hipGetLastError(); /* reported: do not leave it for the next launch's error check */ \
      return GVEC_E_HIP;                                                               \
    }                                                                                  \
  } while (0)

struct gvec_handle {
  gvec_config cfg;
  Variant var;
  int stride, fd, row_dw, mask_dw, mask_bytes, army_dw, maxp;
  hipStream_t stream;
  uint32_t* d_hdr = nullptr;
  uint32_t* d_rows = nullptr;
  uint32_t* d_army16 = nullptr;  // narrow armies (u16 pairs), army_dw / 2 dwords per env
  int32_t* d_army32 = nullptr;   // wide escape (int32), army_dw dwords per env: only envs flagged HF_WIDE use it
  uint32_t* d_legal = nullptr;
  gvec_action* d_actions = nullptr;
  int32_t* d_err = nullptr;
  int32_t* d_status = nullptr;
  uint32_t* d_zeros = nullptr;  // row_dw zero dwords (StepArgs::zeros)
  uint32_t agent_noop = 6554u, agent_half = 19661u;  // gvec_set_agent_mix
  unsigned long long* d_counters = nullptr;  // [6]: before[3], after[3]
  uint32_t* d_snap = nullptr;                // experience snapshots [B][snap_dw] (allocated on first use)
  int snap_dw = 0, record_dw = 0;
  bool record_actions = false;               // per-turn rollouts write the agent's moves into d_actions
  int32_t* d_gym_prev = nullptr;             // [B][3*MAXP] player stats as of the previous gvec_gym_observe
  uint32_t* p_hdr = nullptr;
  uint32_t* p_rows = nullptr;
  uint32_t* p_army16 = nullptr;
  int32_t* p_army32 = nullptr;
  int pool_size = 0;
  uint64_t pool_seed = 0;
  bool legal_valid = false;
  // grow-only device staging for GVEC_MEM_HOST calls: slot i serves the i-th staged argument of a call.
  // Owned by the handle, reused by every call (work on one handle is serialised on its stream), freed by
  // gvec_destroy - the host path allocates nothing in steady state.
  static constexpr int kStageSlots = 24;
  void* stage_ptr[kStageSlots] = {};
  size_t stage_cap[kStageSlots] = {};
  // ---- sharding (gvec_create_sharded) ----
  int env_base = 0;                 // a shard's first env within the sharded batch: keys its agent / pool / map draws
  struct ShardWorker;
  std::vector<std::unique_ptr<ShardWorker>> shards;   // non-empty: this handle owns no device memory, only its shards
  bool sharded() const { return !shards.empty(); }
};

// One worker thread per shard: every call on a sharded handle posts one task per shard and waits for all of them, so the
// shards' host copies, launches and synchronisations run concurrently (a GVEC_MEM_HOST call on a single-device handle ends
// in a stream synchronise; calling the shards one after the other would serialise the devices).  A child handle is only
// ever touched by its own worker: the "not thread-safe per handle" rule holds for every one of them.
struct gvec_handle::ShardWorker {
  gvec_handle* h = nullptr;   // a plain single-device handle
  int begin = 0, n = 0;       // envs [begin, begin + n) of the sharded batch
  std::thread th;
  std::mutex mu;
  std::condition_variable cv;
  std::function<int32_t()> task;
  bool has_task = false, done = false, quit = false;
  int32_t rc = 0;
  std::string err;

  void run() {
    std::unique_lock<std::mutex> lk(mu);
    for (;;) {
      cv.wait(lk, [&] { return has_task || quit; });
      if (quit) return;
      std::function<int32_t()> f = std::move(task);
      has_task = false;
      lk.unlock();
      const int32_t r = f();
      const char* e = gvec_last_error();   // this thread's own message
      lk.lock();
      rc = r;
      err = (r < 0 && e) ? e : "";
      done = true;
      cv.notify_all();
    }
  }
  void post(std::function<int32_t()> f) {
    std::lock_guard<std::mutex> lk(mu);
    task = std::move(f);
    has_task = true;
    done = false;
    cv.notify_all();
  }
  int32_t wait() {
    std::unique_lock<std::mutex> lk(mu);
    cv.wait(lk, [&] { return done; });
    return rc;
  }
  void stop() {
    {
      std::lock_guard<std::mutex> lk(mu);
      quit = true;
      cv.notify_all();
    }
    if (th.joinable()) th.join();
  }
};

namespace {

// One staged argument of the current call: a view of slot `slot` of the handle's grow-only staging.
struct DevBuf {
  gvec_handle* h;
  int slot;
  void* p = nullptr;
  DevBuf(gvec_handle* h_, int slot_) : h(h_), slot(slot_) {}
  hipError_t alloc(size_t bytes) {
    if (bytes < 16) bytes = 16;
    if (h->stage_cap[slot] < bytes) {
      if (h->stage_ptr[slot]) {
        hipError_t e = hipStreamSynchronize(h->stream);  // an earlier call's copy may still read it
        if (e != hipSuccess) return e;
        (void)hipFree(h->stage_ptr[slot]);
        h->stage_ptr[slot] = nullptr;
        h->stage_cap[slot] = 0;
      }
      const size_t cap = bytes + bytes / 4;  // a little headroom: fewer re-allocations while a caller grows
      hipError_t e = hipMalloc(&h->stage_ptr[slot], cap);
      if (e != hipSuccess) return e;
      h->stage_cap[slot] = cap;
    }
    p = h->stage_ptr[slot];
    return hipSuccess;
  }
  template <typename T>
  T* as() const {
    return reinterpret_cast<T*>(p);
  }
};

inline size_t round_up(size_t v, size_t m) { return (v + m - 1) / m * m; }

StepArgs base_args(const gvec_handle* h) {
  StepArgs a;
  memset(&a, 0, sizeof a);
  a.hdr = h->d_hdr;
  a.rows = h->d_rows;
  a.army16 = h->d_army16;
  a.army32 = h->d_army32;
  a.legal = h->d_legal;
  a.zeros = h->d_zeros;
  a.pool_hdr = h->p_hdr;
  a.pool_rows = h->p_rows;
  a.pool_army16 = h->p_army16;
  a.pool_army32 = h->p_army32;
  a.num_envs = h->cfg.num_envs;
  a.fd = h->fd;
  a.row_dw = h->row_dw;
  a.mask_dw = h->mask_dw;
  a.pool_size = h->pool_size;
  a.pstride = h->maxp;
  a.prod_general = h->cfg.prod_general;
  a.prod_city = h->cfg.prod_city;
  a.prod_normal = h->cfg.prod_normal;
  a.interval = h->cfg.normal_growth_interval;
  a.interval_magic = (uint32_t)((0x100000000ull + (uint64_t)a.interval - 1) / (uint64_t)a.interval);
  a.turns = 1;
  a.agent_noop = h->agent_noop;
  a.agent_half = h->agent_half;
  a.pool_seed_lo = (uint32_t)h->pool_seed;
  a.pool_seed_hi = (uint32_t)(h->pool_seed >> 32);
  a.env_base = h->env_base;
  if (h->cfg.auto_reset && h->pool_size > 0) a.flags |= KF_AUTORESET;
  return a;
}

int32_t check_status(gvec_handle* h, const char* what) {
  int32_t st = 0;
  HIPCHK(hipMemcpyAsync(&st, h->d_status, sizeof st, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  if (st != 0) {
    int32_t zero = 0;
    HIPCHK(hipMemcpyAsync(h->d_status, &zero, sizeof zero, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    set_err("%s: input rejected on the device (code %d; -5: board contract - sizes within max_*, owner in [-1, players); "
            "-4: env id out of range)", what, st);
    return st;
  }
  return GVEC_OK;
}

// refresh the internal legal-mask buffer from the resident state
int32_t refresh_legal(gvec_handle* h) {
  StepArgs a = base_args(h);
  HIPCHK(launch_legal(h->var, a, h->stream));
  h->legal_valid = true;
  return GVEC_OK;
}

template <typename T>
int32_t stage_in(gvec_handle* h, DevBuf& buf, const T* src, size_t count, int32_t mem, const T** out) {
  *out = nullptr;
  if (!src) return GVEC_OK;
  if (mem == GVEC_MEM_DEVICE) {
    *out = src;
    return GVEC_OK;
  }
  HIPCHK(buf.alloc(count * sizeof(T)));
  HIPCHK(hipMemcpyAsync(buf.p, src, count * sizeof(T), hipMemcpyHostToDevice, h->stream));
  *out = buf.as<T>();
  return GVEC_OK;
}

template <typename T>
int32_t stage_out(DevBuf& buf, T* dst, size_t count, int32_t mem, T** out) {
  *out = nullptr;
  if (!dst) return GVEC_OK;
  if (mem == GVEC_MEM_DEVICE) {
    *out = dst;
    return GVEC_OK;
  }
  HIPCHK(buf.alloc(count * sizeof(T)));
  *out = buf.as<T>();
  return GVEC_OK;
}

template <typename T>
int32_t copy_out(gvec_handle* h, const DevBuf& buf, T* dst, size_t count, int32_t mem) {
  if (!dst || mem == GVEC_MEM_DEVICE) return GVEC_OK;
  HIPCHK(hipMemcpyAsync(dst, buf.p, count * sizeof(T), hipMemcpyDeviceToHost, h->stream));
  return GVEC_OK;
}

#define RET_IF(x)                \
  do {                           \
    int32_t r__ = (x);           \
    if (r__ != GVEC_OK) return r__; \
  } while (0)

int32_t import_planes(gvec_handle* h, uint32_t* hdr, uint32_t* rows, uint32_t* army16, int32_t* army32, const int32_t* env_ids_dev, int dst_begin,
                      int n, int dst_envs, const gvec_state_view* v /*device pointers*/, bool fresh, bool init) {
  ImportArgs a;
  memset(&a, 0, sizeof a);
  a.hdr = hdr;
  a.rows = rows;
  a.army16 = army16;
  a.army32 = army32;
  a.zeros = h->d_zeros;
  a.env_ids = env_ids_dev;
  a.dst_begin = dst_begin;
  a.n = n;
  a.dst_envs = dst_envs;
  a.s_army = v->army;
  a.s_owner = v->owner;
  a.s_type = v->type;
  a.s_visible = v->visible;
  a.s_listed = v->listed;
  a.s_changed = v->changed;
  a.s_vis_changed = v->vis_changed;
  a.s_turn = v->turn;
  a.s_done = v->done;
  a.s_width = v->width;
  a.s_height = v->height;
  a.s_players = v->players;
  a.s_alive = v->alive;
  a.s_army_count = v->army_count;
  a.s_general_idx = v->general_idx;
  a.stride = h->stride;
  a.max_p = h->maxp;
  a.max_w = h->cfg.max_width;
  a.max_h = h->cfg.max_height;
  a.fd = h->fd;
  a.row_dw = h->row_dw;
  a.fresh = fresh ? 1u : 0u;
  a.init = init ? 1u : 0u;
  a.fog = h->cfg.fog_of_war ? 1u : 0u;
  a.status = h->d_status;
  HIPCHK(launch_import(h->var, a, h->stream));
  if (init) HIPCHK(launch_setup(h->var, a, h->stream));
  return GVEC_OK;
}

int32_t ensure_device() {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) {
    set_err("no HIP device available (%s); this library has no CPU fallback", e == hipSuccess ? "0 devices" : hipGetErrorString(e));
    return GVEC_E_NO_DEVICE;
  }
  return GVEC_OK;
}

// device allocations of a new handle; on failure the caller destroys the handle (which frees what exists)
int32_t allocate_handle(gvec_handle* h, const gvec_config* cfg) {
  const size_t B = (size_t)cfg->num_envs;
  HIPCHK(hipMalloc(&h->d_hdr, B * HDR_DW * 4));
  HIPCHK(hipMalloc(&h->d_rows, B * h->row_dw * 4));
  HIPCHK(hipMalloc(&h->d_army16, B * h->army_dw * 2));
  HIPCHK(hipMalloc(&h->d_army32, B * h->army_dw * 4));
  HIPCHK(hipMalloc(&h->d_legal, B * h->maxp * h->mask_bytes));
  HIPCHK(hipMalloc(&h->d_actions, B * h->maxp * sizeof(gvec_action)));
  HIPCHK(hipMalloc(&h->d_err, B * 4));
  HIPCHK(hipMalloc(&h->d_status, 16));
  HIPCHK(hipMalloc(&h->d_zeros, (size_t)(h->row_dw + 64) * 4));
  HIPCHK(hipMemset(h->d_zeros, 0, (size_t)(h->row_dw + 64) * 4));
  HIPCHK(hipMalloc(&h->d_counters, 6 * sizeof(unsigned long long)));
  HIPCHK(hipMemset(h->d_rows, 0, B * h->row_dw * 4));
  HIPCHK(hipMemset(h->d_army16, 0, B * h->army_dw * 2));
  HIPCHK(hipMemset(h->d_army32, 0, B * h->army_dw * 4));
  HIPCHK(hipMemset(h->d_legal, 0, B * h->maxp * h->mask_bytes));
  HIPCHK(hipMemset(h->d_actions, 0, B * h->maxp * sizeof(gvec_action)));
  HIPCHK(hipMemset(h->d_err, 0, B * 4));
  HIPCHK(hipMemset(h->d_status, 0, 16));
  {  // every slot starts as a finished 1x1 one-player game, so any kernel is safe before gvec_reset
    std::vector<uint32_t> hdr(B * HDR_DW, 0u);
    for (size_t e = 0; e < B; ++e) {
      uint32_t* x = &hdr[e * HDR_DW];
      x[H_DIMS] = 1u | (1u << 8) | (1u << 16) | ((HF_DONE | (cfg->fog_of_war ? HF_FOG : 0u)) << 24);
      x[H_RECIPW] = 65536u;
      for (int p = 0; p < 8; ++p) x[H_GIDX + p] = 0xFFFFFFFFu;
    }
    HIPCHK(hipMemcpy(h->d_hdr, hdr.data(), hdr.size() * 4, hipMemcpyHostToDevice));
  }
  return GVEC_OK;
}

}  // namespace

// =========================================================================================================================
// Sharded handles (gvec_create_sharded): one handle over several devices, SURVEY 8(b) "one handle may span several GPUs".
// Boards are independent, so shard i simply IS envs [begin_i, begin_i + n_i) of the batch (contiguous, sizes differing by
// at most one: the shard_range rule of sharding.py), resident on its own device for the whole run; no call moves board
// state between devices.  Every GVEC_MEM_HOST entry point fans out to the shards with the caller's arrays offset to the
// shard's range, all shards working at once on their own threads and streams.  Because a shard folds its offset into the
// agent / pool / map keys (env_base), the batch plays the same games whatever the number of shards:
// tests/test_hip_sharded.py holds a 3-shard handle against a single-device one bit for bit.
// Entry points that take DEVICE pointers belong to one device: use them on gvec_shard(h, i).
// =========================================================================================================================
namespace sharded {

template <typename F>  // F(gvec_handle* child, int begin, int n) -> int32_t; copied into every shard's task
int32_t fan(gvec_handle* h, F f) {
  for (auto& w : h->shards) {
    gvec_handle::ShardWorker* wp = w.get();
    wp->post([f, wp]() { return f(wp->h, wp->begin, wp->n); });
  }
  int32_t rc = GVEC_OK;
  for (auto& w : h->shards) {
    const int32_t r = w->wait();
    if (r < 0 && rc >= 0) {
      rc = r;
      set_err("shard of envs [%d, %d) on device %d: %s", w->begin, w->begin + w->n, w->h ? w->h->cfg.device : -1, w->err.c_str());
    }
  }
  return rc;
}

// the ordinal of the shard that starts at env `begin` (a handful of shards: linear search)
int ordinal_of(const gvec_handle* h, int begin) {
  for (size_t k = 0; k < h->shards.size(); ++k)
    if (h->shards[k]->begin == begin) return (int)k;
  return 0;
}

int32_t host_only(int32_t mem, const char* what) {
  if (mem == GVEC_MEM_HOST) return GVEC_OK;
  set_err("%s with device pointers on a sharded handle: device memory belongs to one device - call it on gvec_shard(h, i)", what);
  return GVEC_E_INVALID;
}
int32_t unsupported(const char* what) {
  set_err("%s works on one device: call it on gvec_shard(h, i)", what);
  return GVEC_E_INVALID;
}

// the part of a caller's view that covers `skip` envs further on
gvec_state_view offset_view(const gvec_state_view& v, size_t skip, int stride, int maxp) {
  gvec_state_view o = v;
  const size_t t = skip * (size_t)stride, p = skip * (size_t)maxp;
#define GVEC_OFF(field, count) if (o.field) o.field += (count)
  GVEC_OFF(army, t); GVEC_OFF(owner, t); GVEC_OFF(type, t); GVEC_OFF(visible, t); GVEC_OFF(listed, t); GVEC_OFF(changed, t);
  GVEC_OFF(vis_changed, t); GVEC_OFF(turn, skip); GVEC_OFF(done, skip); GVEC_OFF(winner, skip); GVEC_OFF(width, skip);
  GVEC_OFF(height, skip); GVEC_OFF(players, skip); GVEC_OFF(alive, p); GVEC_OFF(army_count, p); GVEC_OFF(tile_count, p);
  GVEC_OFF(general_idx, p);
#undef GVEC_OFF
  return o;
}

// envs [env_begin, env_begin + n) of the batch, split over the shards: f(child, local_begin, count, envs before this piece)
template <typename F>
int32_t fan_range(gvec_handle* h, int32_t env_begin, int32_t n, F f) {
  if (env_begin < 0 || n < 0 || env_begin + n > h->cfg.num_envs) return GVEC_E_RANGE;
  return fan(h, [=](gvec_handle* c, int begin, int cn) -> int32_t {
    const int lo = env_begin > begin ? env_begin : begin, hi = (env_begin + n) < (begin + cn) ? (env_begin + n) : (begin + cn);
    if (hi <= lo) return GVEC_OK;
    return f(c, lo - begin, hi - lo, (size_t)(lo - env_begin));
  });
}

int32_t reset(gvec_handle* h, const int32_t* env_ids, int32_t n, const int32_t* army, const int8_t* owner, const uint8_t* type,
              const int32_t* width, const int32_t* height, const int32_t* players, int32_t mem) {
  RET_IF(host_only(mem, "gvec_reset"));
  const size_t st = (size_t)h->stride;
  if (!env_ids) {
    return fan_range(h, 0, n, [=](gvec_handle* c, int lb, int cnt, size_t skip) {
      return gvec_reset(c, nullptr, cnt, army + skip * st, owner + skip * st, type + skip * st, width + skip, height + skip, players + skip, GVEC_MEM_HOST);
    });
  }
  // explicit ids: every shard gets the rows addressed to it, in the caller's order
  struct Part {
    std::vector<int32_t> ids, w, hh, p, army;
    std::vector<int8_t> owner;
    std::vector<uint8_t> type;
  };
  auto parts = std::make_shared<std::vector<Part>>(h->shards.size());
  for (int i = 0; i < n; ++i) {
    const int e = env_ids[i];
    if (e < 0 || e >= h->cfg.num_envs) return GVEC_E_RANGE;
    size_t k = 0;
    while (e >= h->shards[k]->begin + h->shards[k]->n) ++k;
    Part& P = (*parts)[k];
    P.ids.push_back(e - h->shards[k]->begin);
    P.w.push_back(width[i]);
    P.hh.push_back(height[i]);
    P.p.push_back(players[i]);
    P.army.insert(P.army.end(), army + i * st, army + (i + 1) * st);
    P.owner.insert(P.owner.end(), owner + i * st, owner + (i + 1) * st);
    P.type.insert(P.type.end(), type + i * st, type + (i + 1) * st);
  }
  return fan(h, [parts, h](gvec_handle* c, int begin, int) -> int32_t {
    const Part& P = (*parts)[ordinal_of(h, begin)];
    if (P.ids.empty()) return GVEC_OK;
    return gvec_reset(c, P.ids.data(), (int32_t)P.ids.size(), P.army.data(), P.owner.data(), P.type.data(), P.w.data(), P.hh.data(), P.p.data(),
                      GVEC_MEM_HOST);
  });
}

int32_t gather_records(gvec_handle* h, int32_t local_begin, int32_t n, int32_t env_id_base, int32_t mem, int32_t dst_device, void* dst) {
  if (!dst || n < 0 || local_begin < 0) return GVEC_E_INVALID;
  for (auto& w : h->shards)
    if (local_begin + n > w->n) {
      set_err("gvec_gather_experience_records: envs [%d, %d) of every shard, but a shard holds %d", local_begin, local_begin + n, w->n);
      return GVEC_E_RANGE;
    }
  if (n == 0) return GVEC_OK;
  const size_t rec = (size_t)gvec_experience_record_bytes(h);
  return fan(h, [=](gvec_handle* c, int begin, int) -> int32_t {
    HIPCHK(hipSetDevice(c->cfg.device));
    DevBuf stage(c, gvec_handle::kStageSlots - 1);
    HIPCHK(stage.alloc((size_t)n * rec));
    RET_IF(gvec_experience_records(c, nullptr, GVEC_MEM_DEVICE, local_begin, n, env_id_base + begin, stage.p));
    char* to = reinterpret_cast<char*>(dst) + (size_t)ordinal_of(h, begin) * n * rec;
    if (mem == GVEC_MEM_HOST) HIPCHK(hipMemcpyAsync(to, stage.p, (size_t)n * rec, hipMemcpyDeviceToHost, c->stream));
    else if (dst_device == c->cfg.device) HIPCHK(hipMemcpyAsync(to, stage.p, (size_t)n * rec, hipMemcpyDeviceToDevice, c->stream));
    else HIPCHK(hipMemcpyPeerAsync(to, dst_device, stage.p, c->cfg.device, (size_t)n * rec, c->stream));   // over xGMI
    HIPCHK(hipStreamSynchronize(c->stream));
    return GVEC_OK;
  });
}

}  // namespace sharded

extern "C" {

int32_t gvec_abi_version(void) { return GVEC_ABI_VERSION; }
const char* gvec_last_error(void) { return g_err; }

int32_t gvec_config_default(gvec_config* cfg) {
  if (!cfg) return GVEC_E_INVALID;
  memset(cfg, 0, sizeof *cfg);
  cfg->abi_version = GVEC_ABI_VERSION;
  cfg->num_envs = 1;
  cfg->max_width = 20;
  cfg->max_height = 20;
  cfg->max_players = 2;
  cfg->device = 0;
  cfg->fog_of_war = 1;              // engine_initializer.go:118
  cfg->prod_general = 1;            // config.go:206
  cfg->prod_city = 1;               // config.go:207
  cfg->prod_normal = 1;             // config.go:208
  cfg->normal_growth_interval = 25; // config.go:209
  cfg->auto_reset = 0;
  return GVEC_OK;
}

// the sizes a handle derives from its config (shared by plain and sharded handles)
static bool set_geometry(gvec_handle* h, const gvec_config* cfg) {
  h->cfg = *cfg;
  h->stride = cfg->max_width * cfg->max_height;
  h->maxp = cfg->max_players;
  if (!pick_variant(cfg->max_players, h->stride, &h->var)) {
    set_err("no kernel variant for %d players / %d tiles", cfg->max_players, h->stride);
    return false;
  }
  // dwords per flat bit-plane: 2*nslot-1 or 2*nslot, so that the step kernel can be compiled for it
  h->fd = (h->stride <= 32 * (2 * h->var.nslot - 1)) ? 2 * h->var.nslot - 1 : 2 * h->var.nslot;
  h->row_dw = (int)round_up((size_t)(3 * h->var.maxp + 13) * h->fd, 4);  // Planes<MAXP>::COUNT planes of fd dwords
  h->army_dw = h->var.nslot * 64;
  h->mask_bytes = 16 * h->fd;  // four direction bit-planes of fd dwords per player
  h->mask_dw = h->mask_bytes / 4;
  h->stream = nullptr;
  return true;
}

int32_t gvec_create_sharded(const gvec_config* cfg, const int32_t* devices, int32_t num_devices, gvec_handle** out) {
  if (!cfg || !out || !devices || num_devices < 1 || num_devices > 64) return GVEC_E_INVALID;
  *out = nullptr;
  if (cfg->num_envs < num_devices) {
    set_err("gvec_create_sharded: %d envs over %d shards", cfg->num_envs, num_devices);
    return GVEC_E_INVALID;
  }
  RET_IF(ensure_device());
  int ndev = 0;
  HIPCHK(hipGetDeviceCount(&ndev));
  for (int i = 0; i < num_devices; ++i)
    if (devices[i] < 0 || devices[i] >= ndev) {
      set_err("gvec_create_sharded: device %d of %d", devices[i], ndev);
      return GVEC_E_INVALID;
    }
  gvec_handle* h = new (std::nothrow) gvec_handle();
  if (!h) return GVEC_E_INVALID;
  gvec_config top = *cfg;
  top.device = devices[0];
  if (cfg->abi_version != GVEC_ABI_VERSION || !set_geometry(h, &top)) {
    delete h;
    return GVEC_E_INVALID;
  }
  h->legal_valid = true;
  const int base = cfg->num_envs / num_devices, rem = cfg->num_envs % num_devices;   // sharding.shard_range
  for (int i = 0; i < num_devices; ++i) {
    auto w = std::make_unique<gvec_handle::ShardWorker>();
    w->n = base + (i < rem ? 1 : 0);
    w->begin = i * base + (i < rem ? i : rem);
    gvec_handle::ShardWorker* wp = w.get();
    w->th = std::thread([wp] { wp->run(); });
    h->shards.push_back(std::move(w));
  }
  for (int i = 0; i < num_devices; ++i) {
    gvec_handle::ShardWorker* wp = h->shards[i].get();
    gvec_config c = *cfg;
    c.num_envs = wp->n;
    c.device = devices[i];
    wp->post([wp, c, devices, num_devices, i]() -> int32_t {
      const int32_t rc = gvec_create(&c, &wp->h);
      if (rc != GVEC_OK) return rc;
      wp->h->env_base = wp->begin;
      for (int j = 0; j < num_devices; ++j)   // direct xGMI copies for the record gather (best effort: already on / same device fail harmlessly)
        if (devices[j] != devices[i]) {
          int can = 0;
          if (hipDeviceCanAccessPeer(&can, devices[i], devices[j]) == hipSuccess && can) (void)hipDeviceEnablePeerAccess(devices[j], 0);
          (void)hipGetLastError();
        }
      return GVEC_OK;
    });
  }
  int32_t rc = GVEC_OK;
  for (auto& w : h->shards) {
    const int32_t r = w->wait();
    if (r < 0 && rc >= 0) {
      rc = r;
      set_err("shard of envs [%d, %d): %s", w->begin, w->begin + w->n, w->err.c_str());
    }
  }
  if (rc != GVEC_OK) {
    std::string keep = g_err;
    (void)gvec_destroy(h);
    set_err("%s", keep.c_str());
    return rc;
  }
  *out = h;
  return GVEC_OK;
}

int32_t gvec_num_shards(const gvec_handle* h) { return h ? (int32_t)h->shards.size() : GVEC_E_INVALID; }

int32_t gvec_shard(gvec_handle* h, int32_t i, gvec_handle** child, int32_t* env_begin, int32_t* n, int32_t* device) {
  if (!h || !h->sharded() || i < 0 || i >= (int32_t)h->shards.size()) return GVEC_E_INVALID;
  const auto& w = h->shards[i];
  if (child) *child = w->h;
  if (env_begin) *env_begin = w->begin;
  if (n) *n = w->n;
  if (device) *device = w->h->cfg.device;
  return GVEC_OK;
}

int32_t gvec_gather_experience_records(gvec_handle* h, int32_t shard_env_begin, int32_t n, int32_t env_id_base, int32_t mem, int32_t dst_device,
                                       void* dst) {
  if (!h || !h->sharded()) {
    set_err("gvec_gather_experience_records needs a sharded handle (a plain one writes its records with gvec_experience_records)");
    return GVEC_E_INVALID;
  }
  return sharded::gather_records(h, shard_env_begin, n, env_id_base, mem, dst_device, dst);
}

int32_t gvec_create(const gvec_config* cfg, gvec_handle** out) {
  if (!cfg || !out) return GVEC_E_INVALID;
  *out = nullptr;
  if (cfg->abi_version != GVEC_ABI_VERSION) {
    set_err("abi_version %d != %d", cfg->abi_version, GVEC_ABI_VERSION);
    return GVEC_E_INVALID;
  }
  if (cfg->num_envs < 1 || cfg->max_width < 1 || cfg->max_width > GVEC_MAX_DIM || cfg->max_height < 1 ||
      cfg->max_height > GVEC_MAX_DIM || cfg->max_players < 1 || cfg->max_players > GVEC_MAX_PLAYERS ||
      cfg->normal_growth_interval < 1 || cfg->prod_general < 0 || cfg->prod_city < 0 || cfg->prod_normal < 0 ||
      cfg->prod_general > 0xFFFFFF || cfg->prod_city > 0xFFFFFF || cfg->prod_normal > 0xFFFFFF) {
    set_err("gvec_create: config out of range");
    return GVEC_E_INVALID;
  }
  RET_IF(ensure_device());
  HIPCHK(hipSetDevice(cfg->device));
  gvec_handle* h = new (std::nothrow) gvec_handle();
  if (!h) return GVEC_E_INVALID;
  if (!set_geometry(h, cfg)) {
    delete h;
    return GVEC_E_INVALID;
  }
  {
    const int32_t rc = allocate_handle(h, cfg);
    if (rc != GVEC_OK) {
      (void)gvec_destroy(h);
      return rc;
    }
  }
  h->legal_valid = true;
  *out = h;
  return GVEC_OK;
}

int32_t gvec_destroy(gvec_handle* h) {
  if (!h) return GVEC_E_INVALID;
  if (h->sharded()) {
    for (auto& w : h->shards) {   // each child on the thread that has been driving its device
      gvec_handle::ShardWorker* wp = w.get();
      wp->post([wp]() -> int32_t {
        const int32_t r = wp->h ? gvec_destroy(wp->h) : GVEC_OK;
        wp->h = nullptr;
        return r;
      });
    }
    for (auto& w : h->shards) (void)w->wait();
    for (auto& w : h->shards) w->stop();
    delete h;
    return GVEC_OK;
  }
  (void)hipStreamSynchronize(h->stream);
  void* ptrs[] = {h->d_hdr, h->d_rows, h->d_army16, h->d_army32, h->d_legal, h->d_actions, h->d_err, h->d_status, h->d_zeros, h->d_counters,
                  h->d_snap, h->d_gym_prev, h->p_hdr, h->p_rows, h->p_army16, h->p_army32};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  for (void* p : h->stage_ptr)
    if (p) (void)hipFree(p);
  delete h;
  return GVEC_OK;
}

int32_t gvec_set_stream(gvec_handle* h, void* hip_stream) {
  if (!h) return GVEC_E_INVALID;
  if (h->sharded()) return sharded::unsupported("gvec_set_stream");
  h->stream = reinterpret_cast<hipStream_t>(hip_stream);
  return GVEC_OK;
}
int32_t gvec_synchronize(gvec_handle* h) {
  if (!h) return GVEC_E_INVALID;
  if (h->sharded()) return sharded::fan(h, [](gvec_handle* c, int, int) { return gvec_synchronize(c); });
  HIPCHK(hipStreamSynchronize(h->stream));
  return GVEC_OK;
}

int32_t gvec_host_alloc(uint64_t bytes, void** out) {
  if (!out || bytes == 0) return GVEC_E_INVALID;
  *out = nullptr;
  RET_IF(ensure_device());
  HIPCHK(hipHostMalloc(out, (size_t)bytes, hipHostMallocDefault));
  return GVEC_OK;
}
int32_t gvec_host_free(void* p) {
  if (!p) return GVEC_OK;
  HIPCHK(hipHostFree(p));
  return GVEC_OK;
}

int32_t gvec_num_envs(const gvec_handle* h) { return h ? h->cfg.num_envs : GVEC_E_INVALID; }
int32_t gvec_tile_stride(const gvec_handle* h) { return h ? h->stride : GVEC_E_INVALID; }
int32_t gvec_mask_bytes(const gvec_handle* h) { return h ? h->mask_bytes : GVEC_E_INVALID; }
int64_t gvec_state_bytes_per_env(const gvec_handle* h) {
  return h ? (int64_t)4 * (HDR_DW + h->row_dw + h->army_dw) : (int64_t)GVEC_E_INVALID;
}

int32_t gvec_reset(gvec_handle* h, const int32_t* env_ids, int32_t n, const int32_t* army, const int8_t* owner,
                   const uint8_t* type, const int32_t* width, const int32_t* height, const int32_t* players, int32_t mem) {
  if (!h || n < 0 || !army || !owner || !type || !width || !height || !players) return GVEC_E_INVALID;
  if (n == 0) return GVEC_OK;
  if (h->sharded()) return sharded::reset(h, env_ids, n, army, owner, type, width, height, players, mem);
  if (!env_ids && n > h->cfg.num_envs) return GVEC_E_RANGE;
  if (env_ids && mem == GVEC_MEM_HOST)
    for (int i = 0; i < n; ++i)
      if (env_ids[i] < 0 || env_ids[i] >= h->cfg.num_envs) return GVEC_E_RANGE;
  HIPCHK(hipSetDevice(h->cfg.device));
  DevBuf b_ids(h, 0), b_army(h, 1), b_owner(h, 2), b_type(h, 3), b_w(h, 4), b_h(h, 5), b_p(h, 6);
  const size_t nt = (size_t)n * h->stride;
  gvec_state_view v;
  memset(&v, 0, sizeof v);
  const int32_t* ids = nullptr;
  RET_IF(stage_in(h, b_ids, env_ids, (size_t)n, mem, &ids));
  RET_IF(stage_in(h, b_army, army, nt, mem, (const int32_t**)&v.army));
  RET_IF(stage_in(h, b_owner, owner, nt, mem, (const int8_t**)&v.owner));
  RET_IF(stage_in(h, b_type, type, nt, mem, (const uint8_t**)&v.type));
  RET_IF(stage_in(h, b_w, width, (size_t)n, mem, (const int32_t**)&v.width));
  RET_IF(stage_in(h, b_h, height, (size_t)n, mem, (const int32_t**)&v.height));
  RET_IF(stage_in(h, b_p, players, (size_t)n, mem, (const int32_t**)&v.players));
  RET_IF(import_planes(h, h->d_hdr, h->d_rows, h->d_army16, h->d_army32, ids, 0, n, h->cfg.num_envs, &v, true, true));
  RET_IF(check_status(h, "gvec_reset"));
  return refresh_legal(h);
}

static int32_t generate_into(gvec_handle* h, uint32_t* hdr, uint32_t* rows, uint32_t* army16, int32_t* army32, int count, uint64_t seed,
                             const int32_t* width, const int32_t* height, const int32_t* players, int index_base = 0,
                             const int64_t* go_seeds = nullptr) {
  // Go-seeded boards carry a 607-word generator state each while they are being made: smaller chunks (80 MB of scratch)
  const int chunk = go_seeds ? 16384 : 65536;
  DevBuf b_gs(h, 10), b_gst(h, 11);
  DevBuf b_army(h, 1), b_owner(h, 2), b_type(h, 3), b_w(h, 4), b_h(h, 5), b_p(h, 6), b_iw(h, 7), b_ih(h, 8), b_ip(h, 9);
  const int cn = count < chunk ? count : chunk;
  HIPCHK(b_army.alloc((size_t)cn * h->stride * 4));
  HIPCHK(b_owner.alloc((size_t)cn * h->stride));
  HIPCHK(b_type.alloc((size_t)cn * h->stride));
  HIPCHK(b_w.alloc((size_t)cn * 4));
  HIPCHK(b_h.alloc((size_t)cn * 4));
  HIPCHK(b_p.alloc((size_t)cn * 4));
  if (width) HIPCHK(b_iw.alloc((size_t)cn * 4));
  if (height) HIPCHK(b_ih.alloc((size_t)cn * 4));
  if (players) HIPCHK(b_ip.alloc((size_t)cn * 4));
  if (go_seeds) {
    HIPCHK(b_gs.alloc((size_t)cn * 8));
    HIPCHK(b_gst.alloc((size_t)cn * 607 * 8));
  }
  for (int first = 0; first < count; first += chunk) {
    const int n = (count - first) < chunk ? (count - first) : chunk;
    if (width) HIPCHK(hipMemcpyAsync(b_iw.p, width + first, (size_t)n * 4, hipMemcpyHostToDevice, h->stream));
    if (height) HIPCHK(hipMemcpyAsync(b_ih.p, height + first, (size_t)n * 4, hipMemcpyHostToDevice, h->stream));
    if (players) HIPCHK(hipMemcpyAsync(b_ip.p, players + first, (size_t)n * 4, hipMemcpyHostToDevice, h->stream));
    MapgenArgs m;
    memset(&m, 0, sizeof m);
    m.army = b_army.as<int32_t>();
    m.owner = b_owner.as<int8_t>();
    m.type = b_type.as<uint8_t>();
    m.width = b_w.as<int32_t>();
    m.height = b_h.as<int32_t>();
    m.players = b_p.as<int32_t>();
    m.in_width = width ? b_iw.as<int32_t>() : nullptr;
    m.in_height = height ? b_ih.as<int32_t>() : nullptr;
    m.in_players = players ? b_ip.as<int32_t>() : nullptr;
    m.n = n;
    m.stride = h->stride;
    m.max_w = h->cfg.max_width;
    m.max_h = h->cfg.max_height;
    m.max_p = h->maxp;
    m.first_index = index_base + first;   // board i of a shard is board env_base + i of the batch (pool boards: no offset)
    m.seed_lo = (uint32_t)seed;
    m.seed_hi = (uint32_t)(seed >> 32);
    m.status = h->d_status;
    if (go_seeds) {
      HIPCHK(hipMemcpyAsync(b_gs.p, go_seeds + first, (size_t)n * 8, hipMemcpyHostToDevice, h->stream));
      m.go_seeds = b_gs.as<int64_t>();
      m.go_state = b_gst.as<uint64_t>();
    }
    HIPCHK(launch_mapgen(m, h->stream));
    gvec_state_view v;
    memset(&v, 0, sizeof v);
    v.army = m.army;
    v.owner = m.owner;
    v.type = m.type;
    v.width = m.width;
    v.height = m.height;
    v.players = m.players;
    RET_IF(import_planes(h, hdr, rows, army16, army32, nullptr, first, n, count, &v, true, true));
    HIPCHK(hipStreamSynchronize(h->stream));  // staging is reused by the next chunk
  }
  return check_status(h, "map generation");
}

int32_t gvec_reset_generated(gvec_handle* h, uint64_t seed, const int32_t* width, const int32_t* height,
                             const int32_t* players) {
  if (!h) return GVEC_E_INVALID;
  if (h->sharded())
    return sharded::fan(h, [=](gvec_handle* c, int begin, int) {
      return gvec_reset_generated(c, seed, width ? width + begin : nullptr, height ? height + begin : nullptr, players ? players + begin : nullptr);
    });
  HIPCHK(hipSetDevice(h->cfg.device));
  RET_IF(generate_into(h, h->d_hdr, h->d_rows, h->d_army16, h->d_army32, h->cfg.num_envs, seed, width, height, players, h->env_base));
  return refresh_legal(h);
}

int32_t gvec_reset_go_seeded(gvec_handle* h, const int64_t* seeds, const int32_t* width, const int32_t* height, const int32_t* players) {
  if (!h || !seeds) return GVEC_E_INVALID;
  if (h->sharded())
    return sharded::fan(h, [=](gvec_handle* c, int begin, int) {
      return gvec_reset_go_seeded(c, seeds + begin, width ? width + begin : nullptr, height ? height + begin : nullptr, players ? players + begin : nullptr);
    });
  HIPCHK(hipSetDevice(h->cfg.device));
  RET_IF(generate_into(h, h->d_hdr, h->d_rows, h->d_army16, h->d_army32, h->cfg.num_envs, 0, width, height, players, 0, seeds));
  return refresh_legal(h);
}

int32_t gvec_build_board_pool(gvec_handle* h, int32_t pool_size, uint64_t seed, const int32_t* width, const int32_t* height,
                              const int32_t* players) {
  if (!h || pool_size < 1) return GVEC_E_INVALID;
  if (h->sharded()) {  // every shard keeps its own copy of the same pool (board j is keyed by (seed, j) alone)
    const int32_t rc = sharded::fan(h, [=](gvec_handle* c, int, int) { return gvec_build_board_pool(c, pool_size, seed, width, height, players); });
    if (rc == GVEC_OK) h->pool_size = pool_size;
    return rc;
  }
  HIPCHK(hipSetDevice(h->cfg.device));
  HIPCHK(hipStreamSynchronize(h->stream));
  if (h->p_hdr) (void)hipFree(h->p_hdr);
  if (h->p_rows) (void)hipFree(h->p_rows);
  if (h->p_army16) (void)hipFree(h->p_army16);
  if (h->p_army32) (void)hipFree(h->p_army32);
  h->p_hdr = nullptr;
  h->p_rows = nullptr;
  h->p_army16 = nullptr;
  h->p_army32 = nullptr;
  h->pool_size = 0;
  HIPCHK(hipMalloc(&h->p_hdr, (size_t)pool_size * HDR_DW * 4));
  HIPCHK(hipMalloc(&h->p_rows, (size_t)pool_size * h->row_dw * 4));
  HIPCHK(hipMalloc(&h->p_army16, (size_t)pool_size * h->army_dw * 2));
  HIPCHK(hipMalloc(&h->p_army32, (size_t)pool_size * h->army_dw * 4));
  RET_IF(generate_into(h, h->p_hdr, h->p_rows, h->p_army16, h->p_army32, pool_size, seed, width, height, players));
  h->pool_size = pool_size;
  h->pool_seed = seed;
  return GVEC_OK;
}

int32_t gvec_step(gvec_handle* h, const gvec_action* actions, int32_t* err, uint8_t* legal_bits, int32_t mem) {
  if (!h || !actions) return GVEC_E_INVALID;
  if (h->sharded()) {
    RET_IF(sharded::host_only(mem, "gvec_step"));
    const size_t mp = (size_t)h->maxp, mb = (size_t)h->maxp * h->mask_bytes;
    return sharded::fan(h, [=](gvec_handle* c, int begin, int) {
      return gvec_step(c, actions + begin * mp, err ? err + begin : nullptr, legal_bits ? legal_bits + begin * mb : nullptr, GVEC_MEM_HOST);
    });
  }
  HIPCHK(hipSetDevice(h->cfg.device));
  const size_t B = (size_t)h->cfg.num_envs;
  StepArgs a = base_args(h);
  if (mem == GVEC_MEM_HOST) {
    HIPCHK(hipMemcpyAsync(h->d_actions, actions, B * h->maxp * sizeof(gvec_action), hipMemcpyHostToDevice, h->stream));
    a.actions = h->d_actions;
    a.err = err ? h->d_err : nullptr;
  } else {
    a.actions = actions;
    a.err = err;
  }
  if (legal_bits) {
    // envs that sit the call out (GVEC_ACT_SKIP_ENV) or are frozen write no masks: the buffer must
    // already describe them
    if (!h->legal_valid) RET_IF(refresh_legal(h));
    a.flags |= KF_EMIT | KF_LMVALID;
  }
  HIPCHK(launch_step(h->var, a, h->stream));
  h->legal_valid = legal_bits != nullptr;
  if (mem == GVEC_MEM_HOST) {
    if (err) HIPCHK(hipMemcpyAsync(err, h->d_err, B * 4, hipMemcpyDeviceToHost, h->stream));
    if (legal_bits) HIPCHK(hipMemcpyAsync(legal_bits, h->d_legal, B * h->maxp * h->mask_bytes, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
  } else if (legal_bits && legal_bits != reinterpret_cast<uint8_t*>(h->d_legal)) {
    HIPCHK(hipMemcpyAsync(legal_bits, h->d_legal, B * h->maxp * h->mask_bytes, hipMemcpyDeviceToDevice, h->stream));
  }
  return GVEC_OK;
}

int32_t gvec_legal_mask(gvec_handle* h, uint8_t* legal_bits, int32_t mem) {
  if (!h || !legal_bits) return GVEC_E_INVALID;
  if (h->sharded()) {
    RET_IF(sharded::host_only(mem, "gvec_legal_mask"));
    const size_t mb = (size_t)h->maxp * h->mask_bytes;
    return sharded::fan(h, [=](gvec_handle* c, int begin, int) { return gvec_legal_mask(c, legal_bits + begin * mb, GVEC_MEM_HOST); });
  }
  HIPCHK(hipSetDevice(h->cfg.device));
  if (!h->legal_valid) RET_IF(refresh_legal(h));
  const size_t bytes = (size_t)h->cfg.num_envs * h->maxp * h->mask_bytes;
  if (mem == GVEC_MEM_HOST) {
    HIPCHK(hipMemcpyAsync(legal_bits, h->d_legal, bytes, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
  } else if (legal_bits != reinterpret_cast<uint8_t*>(h->d_legal)) {
    HIPCHK(hipMemcpyAsync(legal_bits, h->d_legal, bytes, hipMemcpyDeviceToDevice, h->stream));
  }
  return GVEC_OK;
}

static int32_t export_range(gvec_handle* h, int32_t env_begin, int32_t n, const gvec_state_view* view, int32_t vis_player,
                            uint8_t* pv_visible, uint8_t* pv_fog, int32_t mem) {
  if (env_begin < 0 || n < 0 || env_begin + n > h->cfg.num_envs) return GVEC_E_RANGE;
  if (n == 0) return GVEC_OK;
  HIPCHK(hipSetDevice(h->cfg.device));
  const size_t nt = (size_t)n * h->stride, np = (size_t)n * h->maxp, ne = (size_t)n;
  static const gvec_state_view kEmpty = {};
  const gvec_state_view* v = view ? view : &kEmpty;
  DevBuf b[19] = {{h, 0}, {h, 1}, {h, 2}, {h, 3}, {h, 4}, {h, 5}, {h, 6}, {h, 7}, {h, 8}, {h, 9}, {h, 10}, {h, 11}, {h, 12},
                  {h, 13}, {h, 14}, {h, 15}, {h, 16}, {h, 17}, {h, 18}};
  ExportArgs a;
  memset(&a, 0, sizeof a);
  a.hdr = h->d_hdr;
  a.rows = h->d_rows;
  a.army16 = h->d_army16;
  a.army32 = h->d_army32;
  a.env_begin = env_begin;
  a.n = n;
  a.stride = h->stride;
  a.max_p = h->maxp;
  a.fd = h->fd;
  a.row_dw = h->row_dw;
  a.vis_player = vis_player;
  RET_IF(stage_out(b[0], v->army, nt, mem, &a.army_out));
  RET_IF(stage_out(b[1], v->owner, nt, mem, &a.owner));
  RET_IF(stage_out(b[2], v->type, nt, mem, &a.type));
  RET_IF(stage_out(b[3], v->visible, nt, mem, &a.visible));
  RET_IF(stage_out(b[4], v->listed, nt, mem, &a.listed));
  RET_IF(stage_out(b[5], v->changed, nt, mem, &a.changed));
  RET_IF(stage_out(b[6], v->vis_changed, nt, mem, &a.vis_changed));
  RET_IF(stage_out(b[7], v->turn, ne, mem, &a.turn));
  RET_IF(stage_out(b[8], v->done, ne, mem, &a.done));
  RET_IF(stage_out(b[9], v->winner, ne, mem, &a.winner));
  RET_IF(stage_out(b[10], v->width, ne, mem, &a.width));
  RET_IF(stage_out(b[11], v->height, ne, mem, &a.height));
  RET_IF(stage_out(b[12], v->players, ne, mem, &a.players));
  RET_IF(stage_out(b[13], v->alive, np, mem, &a.alive));
  RET_IF(stage_out(b[14], v->army_count, np, mem, &a.army_count));
  RET_IF(stage_out(b[15], v->tile_count, np, mem, &a.tile_count));
  RET_IF(stage_out(b[16], v->general_idx, np, mem, &a.general_idx));
  RET_IF(stage_out(b[17], pv_visible, nt, mem, &a.pv_visible));
  RET_IF(stage_out(b[18], pv_fog, nt, mem, &a.pv_fog));
  HIPCHK(launch_export(h->var, a, h->stream));
  RET_IF(copy_out(h, b[0], v->army, nt, mem));
  RET_IF(copy_out(h, b[1], v->owner, nt, mem));
  RET_IF(copy_out(h, b[2], v->type, nt, mem));
  RET_IF(copy_out(h, b[3], v->visible, nt, mem));
  RET_IF(copy_out(h, b[4], v->listed, nt, mem));
  RET_IF(copy_out(h, b[5], v->changed, nt, mem));
  RET_IF(copy_out(h, b[6], v->vis_changed, nt, mem));
  RET_IF(copy_out(h, b[7], v->turn, ne, mem));
  RET_IF(copy_out(h, b[8], v->done, ne, mem));
  RET_IF(copy_out(h, b[9], v->winner, ne, mem));
  RET_IF(copy_out(h, b[10], v->width, ne, mem));
  RET_IF(copy_out(h, b[11], v->height, ne, mem));
  RET_IF(copy_out(h, b[12], v->players, ne, mem));
  RET_IF(copy_out(h, b[13], v->alive, np, mem));
  RET_IF(copy_out(h, b[14], v->army_count, np, mem));
  RET_IF(copy_out(h, b[15], v->tile_count, np, mem));
  RET_IF(copy_out(h, b[16], v->general_idx, np, mem));
  RET_IF(copy_out(h, b[17], pv_visible, nt, mem));
  RET_IF(copy_out(h, b[18], pv_fog, nt, mem));
  if (mem == GVEC_MEM_HOST) HIPCHK(hipStreamSynchronize(h->stream));
  return GVEC_OK;
}

int32_t gvec_player_visibility(gvec_handle* h, int32_t player, uint8_t* visible, uint8_t* fog, int32_t mem) {
  if (!h) return GVEC_E_INVALID;
  if (h->sharded()) {
    RET_IF(sharded::host_only(mem, "gvec_player_visibility"));
    const size_t st = (size_t)h->stride;
    return sharded::fan(h, [=](gvec_handle* c, int begin, int) {
      return gvec_player_visibility(c, player, visible ? visible + begin * st : nullptr, fog ? fog + begin * st : nullptr, GVEC_MEM_HOST);
    });
  }
  return export_range(h, 0, h->cfg.num_envs, nullptr, player, visible, fog, mem);
}

int32_t gvec_read_state(gvec_handle* h, int32_t env_begin, int32_t n, const gvec_state_view* view, int32_t mem) {
  if (!h || !view) return GVEC_E_INVALID;
  if (h->sharded()) {
    RET_IF(sharded::host_only(mem, "gvec_read_state"));
    const gvec_state_view v = *view;
    const int st = h->stride, mp = h->maxp;
    return sharded::fan_range(h, env_begin, n, [=](gvec_handle* c, int lb, int cnt, size_t skip) {
      const gvec_state_view o = sharded::offset_view(v, skip, st, mp);
      return gvec_read_state(c, lb, cnt, &o, GVEC_MEM_HOST);
    });
  }
  return export_range(h, env_begin, n, view, -1, nullptr, nullptr, mem);
}

int32_t gvec_write_state(gvec_handle* h, int32_t env_begin, int32_t n, const gvec_state_view* view, int32_t mem) {
  if (!h || !view) return GVEC_E_INVALID;
  if (h->sharded()) {
    RET_IF(sharded::host_only(mem, "gvec_write_state"));
    const gvec_state_view v = *view;
    const int st = h->stride, mp = h->maxp;
    return sharded::fan_range(h, env_begin, n, [=](gvec_handle* c, int lb, int cnt, size_t skip) {
      const gvec_state_view o = sharded::offset_view(v, skip, st, mp);
      return gvec_write_state(c, lb, cnt, &o, GVEC_MEM_HOST);
    });
  }
  if (env_begin < 0 || n < 0 || env_begin + n > h->cfg.num_envs) return GVEC_E_RANGE;
  if (n == 0) return GVEC_OK;
  if (view->width || view->height || view->players) {
    set_err("gvec_write_state cannot change board dimensions or player count; use gvec_reset");
    return GVEC_E_INVALID;
  }
  HIPCHK(hipSetDevice(h->cfg.device));
  const size_t nt = (size_t)n * h->stride, np = (size_t)n * h->maxp, ne = (size_t)n;
  DevBuf b[12] = {{h, 0}, {h, 1}, {h, 2}, {h, 3}, {h, 4}, {h, 5}, {h, 6}, {h, 7}, {h, 8}, {h, 9}, {h, 10}, {h, 11}};
  gvec_state_view v;
  memset(&v, 0, sizeof v);
  RET_IF(stage_in(h, b[0], view->army, nt, mem, (const int32_t**)&v.army));
  RET_IF(stage_in(h, b[1], view->owner, nt, mem, (const int8_t**)&v.owner));
  RET_IF(stage_in(h, b[2], view->type, nt, mem, (const uint8_t**)&v.type));
  RET_IF(stage_in(h, b[3], view->visible, nt, mem, (const uint8_t**)&v.visible));
  RET_IF(stage_in(h, b[4], view->listed, nt, mem, (const int8_t**)&v.listed));
  RET_IF(stage_in(h, b[5], view->changed, nt, mem, (const uint8_t**)&v.changed));
  RET_IF(stage_in(h, b[6], view->vis_changed, nt, mem, (const uint8_t**)&v.vis_changed));
  RET_IF(stage_in(h, b[7], view->turn, ne, mem, (const int32_t**)&v.turn));
  RET_IF(stage_in(h, b[8], view->done, ne, mem, (const uint8_t**)&v.done));
  RET_IF(stage_in(h, b[9], view->alive, np, mem, (const uint8_t**)&v.alive));
  RET_IF(stage_in(h, b[10], view->army_count, np, mem, (const int32_t**)&v.army_count));
  RET_IF(stage_in(h, b[11], view->general_idx, np, mem, (const int32_t**)&v.general_idx));
  RET_IF(import_planes(h, h->d_hdr, h->d_rows, h->d_army16, h->d_army32, nullptr, env_begin, n, h->cfg.num_envs, &v, false, false));
  RET_IF(check_status(h, "gvec_write_state"));
  return refresh_legal(h);
}

int32_t gvec_rollout(gvec_handle* h, int32_t turns, uint64_t seed, int32_t invalid_permille, int32_t fused,
                     gvec_rollout_stats* stats) {
  if (!h || turns < 0) return GVEC_E_INVALID;
  if (h->sharded()) {
    auto per = std::make_shared<std::vector<gvec_rollout_stats>>(h->shards.size());   // one per shard
    const bool want = stats != nullptr;
    const int32_t rc = sharded::fan(h, [=](gvec_handle* c, int begin, int) {
      return gvec_rollout(c, turns, seed, invalid_permille, fused, want ? &(*per)[sharded::ordinal_of(h, begin)] : nullptr);
    });
    if (stats) {
      memset(stats, 0, sizeof *stats);
      for (const gvec_rollout_stats& p : *per) {
        stats->env_steps += p.env_steps;
        stats->aborted_turns += p.aborted_turns;
        stats->games_finished += p.games_finished;
      }
    }
    return rc;
  }
  HIPCHK(hipSetDevice(h->cfg.device));
  if (stats) {
    HIPCHK(hipMemsetAsync(h->d_counters, 0, 6 * sizeof(unsigned long long), h->stream));
    HIPCHK(launch_counter_sum(h->d_hdr, h->cfg.num_envs, h->d_counters, h->stream));
  }
  StepArgs a = base_args(h);
  a.flags |= KF_AGENT | KF_EMIT;
  a.seed_lo = (uint32_t)seed;
  a.seed_hi = (uint32_t)(seed >> 32);
  a.invalid_permille = invalid_permille;
  if (fused) {
    a.turns = turns;
    if (turns > 0) HIPCHK(launch_rollout(h->var, a, h->stream));
  } else if (turns > 0) {
    if (!h->legal_valid) RET_IF(refresh_legal(h));  // the per-turn agent samples from the mask buffer
    a.turns = 1;
    a.flags |= KF_LMVALID;
    if (h->record_actions) {  // gvec_record_agent_actions: what the agent played, and what the engine said to it
      a.actions_out = h->d_actions;
      a.err = h->d_err;
    }
    for (int k = 0; k < turns; ++k) HIPCHK(launch_step(h->var, a, h->stream));
  }
  if (turns > 0) h->legal_valid = true;
  if (stats) {
    HIPCHK(launch_counter_sum(h->d_hdr, h->cfg.num_envs, h->d_counters + 3, h->stream));
    unsigned long long c[6];
    HIPCHK(hipMemcpyAsync(c, h->d_counters, sizeof c, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    stats->env_steps = (int64_t)(c[3] - c[0]);
    stats->aborted_turns = (int64_t)(c[4] - c[1]);
    stats->games_finished = (int64_t)(c[5] - c[2]);
    stats->reserved = 0;
  }
  return GVEC_OK;
}

int32_t gvec_rollout_range(gvec_handle* h, int32_t env_begin, int32_t n, int32_t turns, uint64_t seed, int32_t invalid_permille) {
  if (!h || turns < 0) return GVEC_E_INVALID;
  if (h->sharded()) return sharded::unsupported("gvec_rollout_range");
  if (env_begin < 0 || n < 0 || env_begin + n > h->cfg.num_envs) return GVEC_E_RANGE;
  if (n == 0 || turns == 0) return GVEC_OK;
  HIPCHK(hipSetDevice(h->cfg.device));
  if (!h->legal_valid) RET_IF(refresh_legal(h));
  // the same launch as gvec_rollout's per-turn path over a slice: every per-env array starts at the slice, and the slice's
  // first env keeps its index in the batch for the agent / pool keys (env_base)
  StepArgs a = base_args(h);
  a.flags |= KF_AGENT | KF_EMIT | KF_LMVALID;
  a.seed_lo = (uint32_t)seed;
  a.seed_hi = (uint32_t)(seed >> 32);
  a.invalid_permille = invalid_permille;
  a.turns = 1;
  const size_t e = (size_t)env_begin;
  a.hdr += e * HDR_DW;
  a.rows += e * h->row_dw;
  a.army16 += e * (h->army_dw / 2);
  a.army32 += e * h->army_dw;
  a.legal += e * h->maxp * h->mask_dw;
  a.num_envs = n;
  a.env_base = h->env_base + env_begin;
  if (h->record_actions) {
    a.actions_out = h->d_actions + e * h->maxp;
    a.err = h->d_err + e;
  }
  for (int k = 0; k < turns; ++k) HIPCHK(launch_step(h->var, a, h->stream));
  return GVEC_OK;
}

int32_t gvec_set_agent_mix(gvec_handle* h, int32_t noop_per_65536, int32_t half_per_65536) {
  if (!h) return GVEC_E_INVALID;
  if (h->sharded()) {
    for (auto& w : h->shards) RET_IF(gvec_set_agent_mix(w->h, noop_per_65536, half_per_65536));  // host-side fields only
    return GVEC_OK;
  }
  if (noop_per_65536 < 0 || noop_per_65536 > 65536 || half_per_65536 < 0 || half_per_65536 > 65536) {
    set_err("gvec_set_agent_mix: thresholds must be in [0, 65536]");
    return GVEC_E_INVALID;
  }
  h->agent_noop = (uint32_t)noop_per_65536;
  h->agent_half = (uint32_t)half_per_65536;
  return GVEC_OK;
}

int32_t gvec_counters(gvec_handle* h, gvec_rollout_stats* out) {
  if (!h || !out) return GVEC_E_INVALID;
  if (h->sharded()) {
    auto per = std::make_shared<std::vector<gvec_rollout_stats>>(h->shards.size());
    const int32_t rc = sharded::fan(h, [=](gvec_handle* c, int begin, int) { return gvec_counters(c, &(*per)[sharded::ordinal_of(h, begin)]); });
    memset(out, 0, sizeof *out);
    for (const gvec_rollout_stats& p : *per) {
      out->env_steps += p.env_steps;
      out->aborted_turns += p.aborted_turns;
      out->games_finished += p.games_finished;
    }
    return rc;
  }
  HIPCHK(hipSetDevice(h->cfg.device));
  HIPCHK(hipMemsetAsync(h->d_counters, 0, 3 * sizeof(unsigned long long), h->stream));
  HIPCHK(launch_counter_sum(h->d_hdr, h->cfg.num_envs, h->d_counters, h->stream));
  unsigned long long c[3];
  HIPCHK(hipMemcpyAsync(c, h->d_counters, sizeof c, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  out->env_steps = (int64_t)c[0];
  out->aborted_turns = (int64_t)c[1];
  out->games_finished = (int64_t)c[2];
  out->reserved = 0;
  return GVEC_OK;
}

int32_t gvec_step_traffic_bytes(const gvec_handle* h, int64_t* out4) {
  if (!h || !out4) return GVEC_E_INVALID;
  static_assert(Planes<4>::MUTABLE == 2 * 4 + 3 && Planes<4>::LST - Planes<4>::MUTABLE == 10 && Planes<4>::COUNT - Planes<4>::LST == 4,
                "gvec_step_traffic_bytes restates the plane block's partition");
  const int64_t fd = h->fd, mp = h->var.maxp;
  const int64_t hdr = HDR_DW * 4;
  const int64_t mut = (2 * mp + 3) * fd * 4;    // Planes<MAXP>::MUTABLE: own, vis, chg, vch, gt1
  const int64_t cst = 10 * fd * 4;              // gen, city, mtn, valid, ncol0, ncolL, ok[4]
  const int64_t lst = mp * fd * 4;              // OwnedTiles planes: only while HF_LDIFF
  const int64_t a16 = (int64_t)h->army_dw * 2, a32 = (int64_t)h->army_dw * 4;
  out4[0] = hdr + mut + cst + a16;
  out4[1] = hdr + mut + a16;
  out4[2] = (int64_t)h->maxp * h->mask_bytes;
  out4[3] = 2 * lst + 2 * a32;
  return GVEC_OK;
}

int32_t gvec_agent_actions(gvec_handle* h, uint64_t seed, int32_t invalid_permille, gvec_action* actions, int32_t mem) {
  if (!h || !actions) return GVEC_E_INVALID;
  if (h->sharded()) {
    RET_IF(sharded::host_only(mem, "gvec_agent_actions"));
    const size_t mp = (size_t)h->maxp;
    return sharded::fan(h, [=](gvec_handle* c, int begin, int) { return gvec_agent_actions(c, seed, invalid_permille, actions + begin * mp, GVEC_MEM_HOST); });
  }
  HIPCHK(hipSetDevice(h->cfg.device));
  StepArgs a = base_args(h);
  a.seed_lo = (uint32_t)seed;
  a.seed_hi = (uint32_t)(seed >> 32);
  a.invalid_permille = invalid_permille;
  a.actions_out = (mem == GVEC_MEM_HOST) ? h->d_actions : actions;
  HIPCHK(launch_agent(h->var, a, h->stream));
  if (mem == GVEC_MEM_HOST) {
    HIPCHK(hipMemcpyAsync(actions, h->d_actions, (size_t)h->cfg.num_envs * h->maxp * sizeof(gvec_action), hipMemcpyDeviceToHost,
                          h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
  }
  return GVEC_OK;
}

static ExperienceArgs exp_args(gvec_handle* h) {
  ExperienceArgs a;
  memset(&a, 0, sizeof a);
  a.hdr = h->d_hdr;
  a.rows = h->d_rows;
  a.army16 = h->d_army16;
  a.army32 = h->d_army32;
  a.snap = h->d_snap;
  a.num_envs = h->cfg.num_envs;
  a.fd = h->fd;
  a.row_dw = h->row_dw;
  a.snap_dw = h->snap_dw;
  a.record_dw = h->record_dw;
  a.pstride = h->maxp;
  a.stride = h->stride;
  a.player = -1;
  return a;
}

static int32_t ensure_snapshots(gvec_handle* h) {
  if (!h->d_snap) {
    experience_layout(h->var, h->fd, &h->snap_dw, &h->record_dw);
    HIPCHK(hipMalloc(&h->d_snap, (size_t)h->cfg.num_envs * h->snap_dw * 4));
    HIPCHK(hipMemset(h->d_snap, 0, (size_t)h->cfg.num_envs * h->snap_dw * 4));
  }
  return GVEC_OK;
}

int32_t gvec_experience_begin_range(gvec_handle* h, int32_t env_begin, int32_t n) {
  if (!h) return GVEC_E_INVALID;
  if (h->sharded())
    return sharded::fan_range(h, env_begin, n, [](gvec_handle* c, int lb, int cnt, size_t) { return gvec_experience_begin_range(c, lb, cnt); });
  if (env_begin < 0 || n < 0 || env_begin + n > h->cfg.num_envs) return GVEC_E_RANGE;
  if (n == 0) return GVEC_OK;
  HIPCHK(hipSetDevice(h->cfg.device));
  RET_IF(ensure_snapshots(h));
  ExperienceArgs a = exp_args(h);
  a.env_begin = env_begin;
  a.num_envs = n;
  HIPCHK(launch_snapshot(h->var, a, h->stream));
  return GVEC_OK;
}

int32_t gvec_experience_begin(gvec_handle* h) {
  if (!h) return GVEC_E_INVALID;
  return gvec_experience_begin_range(h, 0, h->cfg.num_envs);
}

int32_t gvec_experience_rewards(gvec_handle* h, float* rewards, uint8_t* done, int32_t mem) {
  if (!h || !rewards) return GVEC_E_INVALID;
  if (h->sharded()) {
    RET_IF(sharded::host_only(mem, "gvec_experience_rewards"));
    const size_t mp = (size_t)h->maxp;
    return sharded::fan(h, [=](gvec_handle* c, int begin, int) { return gvec_experience_rewards(c, rewards + begin * mp, done ? done + begin : nullptr, GVEC_MEM_HOST); });
  }
  if (!h->d_snap) {
    set_err("gvec_experience_rewards without a preceding gvec_experience_begin");
    return GVEC_E_INVALID;
  }
  HIPCHK(hipSetDevice(h->cfg.device));
  const size_t B = (size_t)h->cfg.num_envs;
  DevBuf br(h, 0), bd(h, 1);
  ExperienceArgs a = exp_args(h);
  RET_IF(stage_out(br, rewards, B * h->maxp, mem, &a.rewards));
  RET_IF(stage_out(bd, done, B, mem, &a.done));
  HIPCHK(launch_rewards(h->var, a, h->stream));
  RET_IF(copy_out(h, br, rewards, B * h->maxp, mem));
  RET_IF(copy_out(h, bd, done, B, mem));
  if (mem == GVEC_MEM_HOST) HIPCHK(hipStreamSynchronize(h->stream));
  return GVEC_OK;
}

int32_t gvec_experience_record_layout(gvec_handle* h, int32_t* out8) {
  if (!h || !out8) return GVEC_E_INVALID;
  int snap_dw = 0, record_dw = 0;
  experience_layout(h->var, h->fd, &snap_dw, &record_dw);
  out8[0] = record_dw;          // dwords per record
  out8[1] = h->var.maxp;        // player slots of the layout (>= max_players)
  out8[2] = h->fd;              // dwords per bit-plane
  out8[3] = h->var.nslot;       // 64-tile army slots
  out8[4] = h->maxp;            // max_players of the handle
  out8[5] = h->stride;          // max_width * max_height
  out8[6] = 0;
  out8[7] = 0;
  return GVEC_OK;
}

int32_t gvec_experience_record_bytes(gvec_handle* h) {
  int32_t l[8];
  const int32_t rc = gvec_experience_record_layout(h, l);
  return rc < 0 ? rc : l[0] * 4;
}

int32_t gvec_experience_records(gvec_handle* h, const gvec_action* actions, int32_t mem, int32_t env_begin, int32_t n,
                                int32_t env_id_base, void* dst_device) {
  if (!h || !dst_device) return GVEC_E_INVALID;
  if (h->sharded()) return sharded::unsupported("gvec_experience_records (a sharded handle collects with gvec_gather_experience_records)");
  if (env_begin < 0 || n < 0 || env_begin + n > h->cfg.num_envs) return GVEC_E_RANGE;
  if (!h->d_snap) {
    set_err("gvec_experience_records without a preceding gvec_experience_begin");
    return GVEC_E_INVALID;
  }
  if (n == 0) return GVEC_OK;
  HIPCHK(hipSetDevice(h->cfg.device));
  ExperienceArgs a = exp_args(h);
  if (!actions) {
    a.actions = h->d_actions;  // the last gvec_step (host mode) / recorded device-agent turn
  } else if (mem == GVEC_MEM_HOST) {
    HIPCHK(hipMemcpyAsync(h->d_actions, actions, (size_t)h->cfg.num_envs * h->maxp * sizeof(gvec_action), hipMemcpyHostToDevice, h->stream));
    a.actions = h->d_actions;
  } else {
    a.actions = actions;
  }
  a.env_begin = env_begin;
  a.num_envs = n;
  a.env_id_base = env_id_base;
  a.records = reinterpret_cast<uint32_t*>(dst_device);
  HIPCHK(launch_experience_records(h->var, a, h->stream));
  return GVEC_OK;
}

int32_t gvec_expand_experience_records(int32_t device, void* hip_stream, const int32_t* layout8, const void* records, int32_t n, float* state,
                                       float* next_state, uint8_t* action_mask, int32_t* meta) {
  if (!layout8 || !records || !state || !next_state || !action_mask || !meta || n < 0) return GVEC_E_INVALID;
  const int rd = layout8[0], mp = layout8[1], fd = layout8[2], ns = layout8[3], stride = layout8[5];
  if (mp < 1 || mp > GVEC_MAX_PLAYERS || fd < 1 || fd > 32 || ns < 1 || ns > 16 || stride < 1 || stride > GVEC_MAX_DIM * GVEC_MAX_DIM ||
      stride > 32 * fd || stride > 64 * ns || rd < 4 + 2 * mp + (8 * mp + 3) * fd + ns * 64 || rd > 4 + 2 * mp + (8 * mp + 3) * fd + ns * 64 + 3) {
    set_err("gvec_expand_experience_records: layout {%d, %d, %d, %d, ., %d} is not one gvec_experience_record_layout produces", rd, mp, fd, ns, stride);
    return GVEC_E_INVALID;
  }
  if (n == 0) return GVEC_OK;
  RET_IF(ensure_device());
  HIPCHK(hipSetDevice(device));
  HIPCHK(launch_expand_records(records, n, layout8, state, next_state, action_mask, meta, reinterpret_cast<hipStream_t>(hip_stream)));
  return GVEC_OK;
}

uint64_t gvec_pool_collect_scratch_bytes(int32_t num_envs) { return num_envs > 0 ? pool_collect_scratch_bytes(num_envs) : 0; }

int32_t gvec_pool_collect(int32_t device, void* hip_stream, const gvec_collect_args* a) {
  if (!a) return GVEC_E_INVALID;
  if (a->num_envs < 1 || a->obs_floats < 1 || a->max_steps_per_episode < 1 || a->result_capacity < 0 || a->capacity < a->num_envs) {
    set_err("gvec_pool_collect: num_envs %d, obs_floats %d, max_steps_per_episode %d, capacity %lld (a step's transitions must fit: >= num_envs), "
            "result_capacity %lld", a->num_envs, a->obs_floats, a->max_steps_per_episode, (long long)a->capacity, (long long)a->result_capacity);
    return GVEC_E_INVALID;
  }
  if (!a->state || !a->next_state || !a->action || !a->reward || !a->terminated || !a->truncated || !a->was_reset || !a->ring_state ||
      !a->ring_next_state || !a->ring_action || !a->ring_reward || !a->ring_done || !a->ring_counters || !a->episode_reward ||
      !a->episode_length || !a->pool_counters || !a->scratch ||
      (a->result_capacity > 0 && (!a->result_reward || !a->result_length || !a->result_worker))) {
    set_err("gvec_pool_collect: a required pointer is NULL (only needs_reset may be)");
    return GVEC_E_INVALID;
  }
  if (reinterpret_cast<uintptr_t>(a->scratch) & 15) {
    set_err("gvec_pool_collect: scratch must be 16-byte aligned");
    return GVEC_E_INVALID;
  }
  RET_IF(ensure_device());
  HIPCHK(hipSetDevice(device));
  HIPCHK(launch_pool_collect(*a, reinterpret_cast<hipStream_t>(hip_stream)));
  return GVEC_OK;
}

int32_t gvec_record_agent_actions(gvec_handle* h, int32_t on) {
  if (!h) return GVEC_E_INVALID;
  if (h->sharded()) {
    for (auto& w : h->shards) RET_IF(gvec_record_agent_actions(w->h, on));
    return GVEC_OK;
  }
  h->record_actions = on != 0;
  return GVEC_OK;
}

static int32_t gym_observe_impl(gvec_handle* h, int32_t player, const int64_t* turn_count, int32_t max_turns, float* obs, uint8_t* mask,
                                double* reward, uint8_t* done, int8_t* winner, const uint8_t* resetting, const uint8_t* played, int64_t* turn_io,
                                int64_t* turn_out, uint8_t* terminated, uint8_t* truncated, uint8_t* needs_reset);

int32_t gvec_gym_observe(gvec_handle* h, int32_t player, const int64_t* turn_count, int32_t max_turns, float* obs, uint8_t* mask,
                         double* reward, uint8_t* done, int8_t* winner) {
  return gym_observe_impl(h, player, turn_count, max_turns, obs, mask, reward, done, winner, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                          nullptr);
}

int32_t gvec_gym_finish_step(gvec_handle* h, int32_t player, int64_t* turn_count, int32_t max_turns, const uint8_t* resetting,
                             const uint8_t* played, float* obs, uint8_t* mask, double* reward, uint8_t* terminated, uint8_t* truncated,
                             int8_t* winner, uint8_t* needs_reset, int64_t* turn_out) {
  if (!resetting || !played) return GVEC_E_INVALID;
  return gym_observe_impl(h, player, turn_count, max_turns, obs, mask, reward, nullptr, winner, resetting, played, turn_count, turn_out, terminated,
                          truncated, needs_reset);
}

static int32_t gym_observe_impl(gvec_handle* h, int32_t player, const int64_t* turn_count, int32_t max_turns, float* obs, uint8_t* mask,
                                double* reward, uint8_t* done, int8_t* winner, const uint8_t* resetting, const uint8_t* played, int64_t* turn_io,
                                int64_t* turn_out, uint8_t* terminated, uint8_t* truncated, uint8_t* needs_reset) {
  if (!h || !turn_count || !obs || !mask || player < 0 || player >= h->maxp || max_turns < 1) return GVEC_E_INVALID;
  if (h->sharded()) return sharded::unsupported("gvec_gym_observe / gvec_gym_finish_step");
  HIPCHK(hipSetDevice(h->cfg.device));
  if (!h->d_gym_prev) {
    HIPCHK(hipMalloc(&h->d_gym_prev, (size_t)h->cfg.num_envs * 3 * h->var.maxp * 4));
    HIPCHK(hipMemsetAsync(h->d_gym_prev, 0, (size_t)h->cfg.num_envs * 3 * h->var.maxp * 4, h->stream));
  }
  GymArgs a;
  memset(&a, 0, sizeof a);
  a.hdr = h->d_hdr;
  a.rows = h->d_rows;
  a.army16 = h->d_army16;
  a.army32 = h->d_army32;
  a.turn_count = turn_count;
  a.obs = obs;
  a.mask = mask;
  a.reward = reward;
  a.done = done;
  a.winner = winner;
  a.prev_stats = h->d_gym_prev;
  a.resetting = resetting;
  a.played = played;
  a.turn_io = turn_io;
  a.turn_out = turn_out;
  a.terminated = terminated;
  a.truncated = truncated;
  a.needs_reset = needs_reset;
  a.num_envs = h->cfg.num_envs;
  a.fd = h->fd;
  a.row_dw = h->row_dw;
  a.stride = h->stride;
  a.player = player;
  a.max_turns = max_turns;
  HIPCHK(launch_gym_observe(h->var, a, h->stream));
  return GVEC_OK;
}

int32_t gvec_gym_actions(gvec_handle* h, int32_t player, const int64_t* gym_actions, const uint8_t* mask, const uint8_t* resetting,
                         gvec_action* actions, uint8_t* played, uint8_t* invalid, uint8_t* error) {
  if (!h || !gym_actions || !mask || !actions || player < 0 || player >= h->maxp) return GVEC_E_INVALID;
  if (h->sharded()) return sharded::unsupported("gvec_gym_actions");
  HIPCHK(hipSetDevice(h->cfg.device));
  GymActArgs a;
  memset(&a, 0, sizeof a);
  a.hdr = h->d_hdr;
  a.gym_actions = gym_actions;
  a.mask = mask;
  a.resetting = resetting;
  a.actions = actions;
  a.played = played;
  a.invalid = invalid;
  a.error = error;
  a.num_envs = h->cfg.num_envs;
  a.stride = h->stride;
  a.pstride = h->maxp;
  a.player = player;
  HIPCHK(launch_gym_actions(a, h->stream));
  return GVEC_OK;
}

int32_t gvec_gym_step(gvec_handle* h, int32_t player, uint64_t agent_seed, const int64_t* gym_actions, const uint8_t* resetting,
                      int64_t* turn_count, int32_t max_turns, float* obs, uint8_t* mask, double* reward, uint8_t* terminated,
                      uint8_t* truncated, int8_t* winner, uint8_t* needs_reset, int64_t* turn_out, uint8_t* played, uint8_t* invalid,
                      uint8_t* error) {
  if (!h || !gym_actions || !resetting || !turn_count || !obs || !mask || player < 0 || player >= h->maxp || max_turns < 1) return GVEC_E_INVALID;
  if (h->sharded()) return sharded::unsupported("gvec_gym_step");
  if (!(h->cfg.auto_reset && h->pool_size > 0)) {
    set_err("gvec_gym_step needs auto_reset and a board pool (gvec_build_board_pool): episodes end by re-dealing");
    return GVEC_E_INVALID;
  }
  HIPCHK(hipSetDevice(h->cfg.device));
  if (!h->d_gym_prev) {
    HIPCHK(hipMalloc(&h->d_gym_prev, (size_t)h->cfg.num_envs * 3 * h->var.maxp * 4));
    HIPCHK(hipMemsetAsync(h->d_gym_prev, 0, (size_t)h->cfg.num_envs * 3 * h->var.maxp * 4, h->stream));
  }
  StepArgs a = base_args(h);
  a.seed_lo = (uint32_t)agent_seed;
  a.seed_hi = (uint32_t)(agent_seed >> 32);
  a.invalid_permille = 0;
  GymStepArgs g;
  memset(&g, 0, sizeof g);
  g.gym_actions = gym_actions;
  g.resetting = resetting;
  g.turn_io = turn_count;
  g.turn_out = turn_out;
  g.obs = obs;
  g.mask = mask;
  g.reward = reward;
  g.terminated = terminated;
  g.truncated = truncated;
  g.winner = winner;
  g.needs_reset = needs_reset;
  g.played = played;
  g.invalid = invalid;
  g.error = error;
  g.prev_stats = h->d_gym_prev;
  g.stride = h->stride;
  g.player = player;
  g.max_turns = max_turns;
  HIPCHK(launch_gym_step(h->var, a, g, h->stream));
  h->legal_valid = false;  // the engine's own mask buffer was not refreshed
  return GVEC_OK;
}

int32_t gvec_stream_delta_cap(const gvec_handle* h) { return h ? (h->stride / 5 > 1 ? h->stride / 5 : 1) : GVEC_E_INVALID; }

int32_t gvec_stream_deltas(gvec_handle* h, int32_t player, uint8_t* kind, int32_t* count, uint64_t* updates, int32_t mem) {
  if (!h || !kind || !count || !updates || player < 0 || player >= h->maxp) return GVEC_E_INVALID;
  const int cap = gvec_stream_delta_cap(h);
  if (h->sharded()) {
    RET_IF(sharded::host_only(mem, "gvec_stream_deltas"));
    return sharded::fan(h, [=](gvec_handle* c, int begin, int) {
      return gvec_stream_deltas(c, player, kind + begin, count + begin, updates + (size_t)begin * cap, GVEC_MEM_HOST);
    });
  }
  HIPCHK(hipSetDevice(h->cfg.device));
  const size_t B = (size_t)h->cfg.num_envs;
  DevBuf bk(h, 0), bc(h, 1), bu(h, 2);
  StreamDeltaArgs a;
  memset(&a, 0, sizeof a);
  a.hdr = h->d_hdr;
  a.rows = h->d_rows;
  a.army16 = h->d_army16;
  a.army32 = h->d_army32;
  a.num_envs = h->cfg.num_envs;
  a.fd = h->fd;
  a.row_dw = h->row_dw;
  a.player = player;
  a.cap = cap;
  unsigned long long* du = nullptr;
  RET_IF(stage_out(bk, kind, B, mem, &a.kind));
  RET_IF(stage_out(bc, count, B, mem, &a.count));
  RET_IF(stage_out(bu, reinterpret_cast<unsigned long long*>(updates), B * cap, mem, &du));
  a.updates = du;
  HIPCHK(launch_stream_deltas(h->var, a, h->stream));
  RET_IF(copy_out(h, bk, kind, B, mem));
  RET_IF(copy_out(h, bc, count, B, mem));
  RET_IF(copy_out(h, bu, reinterpret_cast<unsigned long long*>(updates), B * cap, mem));
  if (mem == GVEC_MEM_HOST) HIPCHK(hipStreamSynchronize(h->stream));
  return GVEC_OK;
}

int32_t gvec_stream_deltas_packed(gvec_handle* h, int32_t player, int32_t full_tiles, uint8_t* kind, int64_t* offset, uint64_t* updates,
                                  int64_t capacity, int64_t* total) {
  if (!h || !kind || !offset || !updates || !total || capacity < 0 || player < 0 || player >= h->maxp) return GVEC_E_INVALID;
  if (h->sharded()) return sharded::unsupported("gvec_stream_deltas_packed");
  HIPCHK(hipSetDevice(h->cfg.device));
  const int cap = full_tiles ? h->stride : gvec_stream_delta_cap(h);   // rows long enough for a whole board when asked for
  const size_t B = (size_t)h->cfg.num_envs;
  DevBuf bk(h, 0), bc(h, 1), bu(h, 2), bo(h, 3), bp(h, 4);
  HIPCHK(bk.alloc(B));
  HIPCHK(bc.alloc(B * 4));
  HIPCHK(bu.alloc(B * cap * 8));
  HIPCHK(bo.alloc((B + 1) * 8));
  HIPCHK(bp.alloc(B * cap * 8));
  StreamDeltaArgs a;
  memset(&a, 0, sizeof a);
  a.hdr = h->d_hdr;
  a.rows = h->d_rows;
  a.army16 = h->d_army16;
  a.army32 = h->d_army32;
  a.num_envs = h->cfg.num_envs;
  a.fd = h->fd;
  a.row_dw = h->row_dw;
  a.player = player;
  a.cap = cap;
  a.full_tiles = full_tiles ? 1 : 0;
  a.kind = bk.as<uint8_t>();
  a.count = bc.as<int32_t>();
  a.updates = bu.as<unsigned long long>();
  HIPCHK(launch_stream_deltas(h->var, a, h->stream));
  HIPCHK(launch_pack_updates(bu.as<unsigned long long>(), bc.as<int32_t>(), bo.as<long long>(), bp.as<unsigned long long>(), h->cfg.num_envs, cap,
                             (long long)(B * cap), h->stream));
  HIPCHK(hipMemcpyAsync(kind, bk.p, B, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipMemcpyAsync(offset, bo.p, (B + 1) * 8, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  *total = offset[B];
  if (*total > capacity) {
    set_err("gvec_stream_deltas_packed: %lld updates, room for %lld", (long long)*total, (long long)capacity);
    return GVEC_E_RANGE;
  }
  if (*total > 0) {
    HIPCHK(hipMemcpyAsync(updates, bp.p, (size_t)*total * 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
  }
  return GVEC_OK;
}

int32_t gvec_observe(gvec_handle* h, int32_t player, float* out, int32_t mem) {
  if (!h || !out || player < -1 || player >= h->maxp) return GVEC_E_INVALID;
  if (h->sharded()) {
    RET_IF(sharded::host_only(mem, "gvec_observe"));
    const size_t per = (size_t)(player < 0 ? h->maxp : 1) * 9 * h->stride;
    return sharded::fan(h, [=](gvec_handle* c, int begin, int) { return gvec_observe(c, player, out + begin * per, GVEC_MEM_HOST); });
  }
  HIPCHK(hipSetDevice(h->cfg.device));
  const size_t count = (size_t)h->cfg.num_envs * (player < 0 ? h->maxp : 1) * 9 * h->stride;
  DevBuf bo(h, 0);
  ExperienceArgs a = exp_args(h);
  a.player = player;
  RET_IF(stage_out(bo, out, count, mem, &a.obs));
  HIPCHK(launch_observe(h->var, a, h->stream));
  RET_IF(copy_out(h, bo, out, count, mem));
  if (mem == GVEC_MEM_HOST) HIPCHK(hipStreamSynchronize(h->stream));
  return GVEC_OK;
}

int32_t gvec_serializer_mask(gvec_handle* h, uint8_t* bits, int32_t mem) {
  if (!h || !bits) return GVEC_E_INVALID;
  if (h->sharded()) {
    RET_IF(sharded::host_only(mem, "gvec_serializer_mask"));
    const size_t mb = (size_t)h->maxp * h->mask_bytes;
    return sharded::fan(h, [=](gvec_handle* c, int begin, int) { return gvec_serializer_mask(c, bits + begin * mb, GVEC_MEM_HOST); });
  }
  HIPCHK(hipSetDevice(h->cfg.device));
  const size_t bytes = (size_t)h->cfg.num_envs * h->maxp * h->mask_bytes;
  DevBuf bb(h, 0);
  StepArgs a = base_args(h);
  uint8_t* dst = nullptr;
  RET_IF(stage_out(bb, bits, bytes, mem, &dst));
  a.legal = reinterpret_cast<uint32_t*>(dst);
  HIPCHK(launch_serializer_mask(h->var, a, h->stream));
  RET_IF(copy_out(h, bb, bits, bytes, mem));
  if (mem == GVEC_MEM_HOST) HIPCHK(hipStreamSynchronize(h->stream));
  return GVEC_OK;
}

static RecordArgs record_args(gvec_handle* h, int32_t env_begin, int32_t n, void* slab) {
  RecordArgs a;
  memset(&a, 0, sizeof a);
  a.hdr = h->d_hdr;
  a.rows = h->d_rows;
  a.army16 = h->d_army16;
  a.army32 = h->d_army32;
  char* d = reinterpret_cast<char*>(slab);
  const size_t hb = (size_t)n * HDR_DW * 4, rb = (size_t)n * h->row_dw * 4;
  a.rec_hdr = reinterpret_cast<uint32_t*>(d);
  a.rec_rows = reinterpret_cast<uint32_t*>(d + hb);
  a.rec_army = reinterpret_cast<int32_t*>(d + hb + rb);
  a.env_begin = env_begin;
  a.n = n;
  a.fd = h->fd;
  a.row_dw = h->row_dw;
  a.max_w = h->cfg.max_width;
  a.max_h = h->cfg.max_height;
  a.max_p = h->maxp;
  a.status = h->d_status;
  return a;
}

int32_t gvec_export_records(gvec_handle* h, int32_t env_begin, int32_t n, void* dst_device) {
  if (!h || !dst_device) return GVEC_E_INVALID;
  if (h->sharded()) return sharded::unsupported("gvec_export_records");
  if (env_begin < 0 || n < 0 || env_begin + n > h->cfg.num_envs) return GVEC_E_RANGE;
  if (n == 0) return GVEC_OK;
  HIPCHK(hipSetDevice(h->cfg.device));
  HIPCHK(launch_records(h->var, record_args(h, env_begin, n, dst_device), false, h->stream));
  return GVEC_OK;
}

int32_t gvec_import_records(gvec_handle* h, int32_t env_begin, int32_t n, const void* src_device) {
  if (!h || !src_device) return GVEC_E_INVALID;
  if (h->sharded()) return sharded::unsupported("gvec_import_records");
  if (env_begin < 0 || n < 0 || env_begin + n > h->cfg.num_envs) return GVEC_E_RANGE;
  if (n == 0) return GVEC_OK;
  HIPCHK(hipSetDevice(h->cfg.device));
  HIPCHK(launch_records(h->var, record_args(h, env_begin, n, const_cast<void*>(src_device)), true, h->stream));
  h->legal_valid = false;
  // every record's header was checked on the device before anything was taken from it
  return check_status(h, "gvec_import_records");
}

int32_t gvec_read_buffer(gvec_handle* h, int32_t which, uint64_t byte_offset, uint64_t bytes, void* host_dst) {
  if (!h || !host_dst) return GVEC_E_INVALID;
  if (h->sharded()) return sharded::unsupported("gvec_read_buffer");
  const size_t B = (size_t)h->cfg.num_envs;
  size_t total = 0;
  const void* base = nullptr;
  switch (which) {
    case GVEC_BUF_HEADER: base = h->d_hdr; total = B * HDR_DW * 4; break;
    case GVEC_BUF_LEGAL: base = h->d_legal; total = B * h->maxp * h->mask_bytes; break;
    case GVEC_BUF_ACTIONS: base = h->d_actions; total = B * h->maxp * sizeof(gvec_action); break;
    case GVEC_BUF_ERR: base = h->d_err; total = B * 4; break;
    default:
      set_err("gvec_read_buffer: buffer %d is not one of header / legal masks / actions / err", which);
      return GVEC_E_INVALID;
  }
  if (byte_offset > total || bytes > total - byte_offset) return GVEC_E_RANGE;
  if (bytes == 0) return GVEC_OK;
  HIPCHK(hipSetDevice(h->cfg.device));
  HIPCHK(hipMemcpyAsync(host_dst, reinterpret_cast<const char*>(base) + byte_offset, (size_t)bytes, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return GVEC_OK;
}

void* gvec_device_buffer(gvec_handle* h, int32_t which) {
  if (!h) return nullptr;
  if (h->sharded()) return nullptr;   // device memory belongs to one device: gvec_device_buffer(gvec_shard(h, i), which)
  switch (which) {
    case 0: return h->d_hdr;
    case 1: return h->d_rows;
    case 2: return h->d_army16;
    case 6: return h->d_army32;
    case 3: return h->d_legal;
    case 4: return h->d_actions;
    case 5: return h->d_err;
    default: return nullptr;
  }
}

int32_t gvec_selftest(int32_t device) {
  RET_IF(ensure_device());
  HIPCHK(hipSetDevice(device));
  int32_t* d = nullptr;
  HIPCHK(hipMalloc(&d, 16));
  HIPCHK(hipMemset(d, 0xFF, 16));
  hipError_t e = launch_selftest(d, nullptr);
  int32_t out = -1;
  if (e == hipSuccess) e = hipMemcpy(&out, d, 4, hipMemcpyDeviceToHost);
  (void)hipFree(d);
  if (e != hipSuccess) {
    set_err("selftest: %s", hipGetErrorString(e));
    return GVEC_E_HIP;
  }
  if (out != 0) set_err("wave-primitive self-test failed: lane %d check %d", out / 16, out % 16);
  return out;
}

}  // extern "C"
