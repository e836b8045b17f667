// gvec_device.hpp — CDNA4 (gfx950) device code of the batched Generals.io turn engine.
//
// One 64-lane wavefront owns one board ("env").  The board lives in registers in two
// layouts at once (DESIGN.md "Data layout"):
//
//   * ROW domain  — lane y holds row y of every bit-plane as one u32 (W <= 32 bits):
//       own[p]  tile.Owner == p            (core/board.go:8)
//       lst[p]  tile in Players[p].OwnedTiles (game/state.go:12; SURVEY H6 "plane L")
//       vis[p]  bit p of tile.VisibleBitfield (core/board.go:11)
//       chg     GameState.ChangedTiles, vch GameState.VisibilityChangedTiles (state.go:29,33)
//       gen/city/mtn  tile.Type one-hot     (core/board.go:20-26)
//     3x3 / 5x5 stencils, ownership algebra and the legal-move predicate are a handful
//     of shifts, DPP wave shifts and ANDs here.
//   * TILE domain — lane l, slot s holds Tile.Army of tile t = 64*s + l (int32, H12).
//
// Row planes are staged HBM -> LDS -> registers with flat 16-byte coalesced accesses;
// armies are loaded straight into registers, 256 B per wave instruction.
// Everything is integer / bit work: no MFMA anywhere (HBM-roofline kernel).
//
// Every routine cites the Go function it reproduces (paths relative to
// /root/reference/internal/game/).  Semantics are the plane re-statement derived in
// SURVEY.md section 8 (H1-H12); tests/ checks them bit-for-bit against oracle/.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/generals_vec.h"

namespace gvec {

// ---- resident record layout (per env) -------------------------------------------------
// hdr  : HDR_DW u32          rows : M*HS u32, plane-major [m][y]     army : NSLOT*64 i32
constexpr int HDR_DW = 24;
enum : int {
  H_TURN = 0,     // GameState.Turn
  H_DIMS = 1,     // W | H<<8 | P<<16 | flags<<24   (flags: bit0 Engine.gameOver, bit1 FogOfWarEnabled)
  H_STATUS = 2,   // alive bits | (winner+1)<<8 | last err<<16
  H_EPISODE = 3,  // re-deal counter (auto-reset)
  H_ARMYCNT = 4,  // [8] Player.ArmyCount
  H_GIDX = 12,    // [8] Player.GeneralIdx
  H_RECIPW = 20,  // ceil(65536 / W): exact t / W for t < 1024 (see tile_coords)
  H_CNT_STEPS = 21,   // lifetime counters of this env slot (rollout statistics)
  H_CNT_ABORT = 22,
  H_CNT_DONE = 23
};
constexpr uint32_t HF_DONE = 1u, HF_FOG = 2u;

// plane order inside the rows block; the last three never change after reset
template <int MAXP>
struct Planes {
  static constexpr int OWN = 0, LST = MAXP, VIS = 2 * MAXP, CHG = 3 * MAXP, VCH = 3 * MAXP + 1, GEN = 3 * MAXP + 2,
                       CITY = 3 * MAXP + 3, MTN = 3 * MAXP + 4, COUNT = 3 * MAXP + 5, MUTABLE = 3 * MAXP + 2;
};

constexpr uint32_t KF_AGENT = 1u;      // sample actions on device instead of reading them
constexpr uint32_t KF_EMIT = 2u;       // write legal-action masks
constexpr uint32_t KF_AUTORESET = 4u;  // re-deal finished envs from the pool
constexpr uint32_t KF_LMVALID = 16u;   // args.legal already holds the masks of the current state

struct StepArgs {
  uint32_t* hdr;
  uint32_t* rows;
  int32_t* army;
  const gvec_action* actions;  // [B][MAXP] (ignored with KF_AGENT)
  gvec_action* actions_out;    // optional: where the agent records what it played
  int32_t* err;                // [B] or null
  uint32_t* legal;             // [B][MAXP][mask_dw]
  const uint32_t* pool_hdr;
  const uint32_t* pool_rows;
  const int32_t* pool_army;
  int32_t num_envs, hs, row_dw, mask_dw, pool_size;
  int32_t pstride;  // players per env in actions / legal buffers (gvec_config.max_players)
  int32_t prod_general, prod_city, prod_normal, interval;
  int32_t turns, invalid_permille;
  uint32_t flags, seed_lo, seed_hi, pool_seed_lo, pool_seed_hi;
};

// ---- wave primitives --------------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63u); }

template <int CTRL, int RM = 0xf, int BM = 0xf>
__device__ __forceinline__ uint32_t dpp0(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, RM, BM, false);
}
// lane y receives lane y-1 (row above); lane 0 receives 0     [DPP wave_shr:1]
__device__ __forceinline__ uint32_t from_above(uint32_t v) { return dpp0<0x138>(v); }
// lane y receives lane y+1 (row below); lane 63 receives 0    [DPP wave_shl:1]
__device__ __forceinline__ uint32_t from_below(uint32_t v) { return dpp0<0x130>(v); }

// inclusive prefix sum over the 64 lanes (row_shr 1/2/4/8, then row_bcast 15/31)
__device__ __forceinline__ uint32_t wave_scan_add(uint32_t v) {
  v += dpp0<0x111>(v);
  v += dpp0<0x112>(v);
  v += dpp0<0x114>(v);
  v += dpp0<0x118>(v);
  v += dpp0<0x142, 0xa>(v);
  v += dpp0<0x143, 0xc>(v);
  return v;
}
__device__ __forceinline__ uint32_t rdlane(uint32_t v, int lane) {
  return (uint32_t)__builtin_amdgcn_readlane((int)v, lane);
}
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ uint32_t uniu(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ uint32_t wave_sum(uint32_t v) { return rdlane(wave_scan_add(v), 63); }
__device__ __forceinline__ bool wave_any(bool c) { return __builtin_amdgcn_ballot_w64(c) != 0ull; }
// NOTE: ds_bpermute / DPP read 0 from lanes that are masked off in EXEC.  Every cross-lane
// helper below must therefore be called under wave-uniform control flow only.
__device__ __forceinline__ uint32_t bperm(int byte_addr, uint32_t v) {
  return (uint32_t)__builtin_amdgcn_ds_bpermute(byte_addr, (int)v);
}
// LDS traffic inside one wave needs no s_barrier (DS ops of a wave execute in order); this
// only stops the compiler from reordering the accesses of different lanes.
__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// ---- the build's counter RNG (DESIGN.md "Synthetic inputs"; mirrored in oracle/) --------
__device__ __forceinline__ uint32_t fmix32(uint32_t h) {
  h ^= h >> 16;
  h *= 0x85EBCA6Bu;
  h ^= h >> 13;
  h *= 0xC2B2AE35u;
  h ^= h >> 16;
  return h;
}
__device__ __forceinline__ uint32_t env_key(uint32_t lo, uint32_t hi, uint32_t env) {
  return fmix32(fmix32(lo ^ 0x9E3779B9u) + hi * 0x85EBCA77u + env * 0xC2B2AE3Du + 0x27D4EB2Fu);
}

// spread bit i of the low byte to bit 4*i
__device__ __forceinline__ uint32_t spread4(uint32_t x) {
  x &= 0xFFu;
  x = (x | (x << 12)) & 0x000F000Fu;
  x = (x | (x << 6)) & 0x03030303u;
  x = (x | (x << 3)) & 0x11111111u;
  return x;
}

// =========================================================================================
template <int MAXP, int NSLOT>
struct Board {
  using PL = Planes<MAXP>;
  static constexpr int MPASS = (NSLOT > 8) ? 2 : 1;  // legal-mask dwords per lane (8 tiles each)

  // row domain
  uint32_t own[MAXP], lst[MAXP], vis[MAXP], chg, vch, gen, city, mtn;
  uint32_t rowmask;  // in-board bits of this lane's row (0 for y >= H)
  // tile domain
  int32_t army[NSLOT];
  int ysel[NSLOT];  // 4 * min(y, 63) of tile 64*s + lane   (byte address for ds_bpermute)
  int xsh[NSLOT];   // x of that tile
  // flat-byte domain: lane j (+64*pass) assembles bits [8j, 8j+8) of a plane's row-major bit string
  int fb_ysel[MPASS], fb_x0[MPASS];
  // wave-uniform
  int W, H, P, N, turn, episode, recipW;
  uint32_t alive, hflags, last_err, cnt_steps, cnt_abort, cnt_done;
  int winner;
  int army_count[MAXP], gidx[MAXP];

  // ---------------------------------------------------------------------------------------
  __device__ __forceinline__ void tile_coords() {
    const int lane = lane_id();
#pragma unroll
    for (int s = 0; s < NSLOT; ++s) {
      int t = 64 * s + lane;
      int y = (t * recipW) >> 16;  // == t / W for t < 1024, W <= 32 (error term < 1/64 < 1/W)
      xsh[s] = t - y * W;
      ysel[s] = 4 * (y < 63 ? y : 63);  // lane 63 never holds a board row (H <= 32)
    }
#pragma unroll
    for (int k = 0; k < MPASS; ++k) {
      int f = 8 * (lane + 64 * k);
      int y = (f * recipW) >> 16;
      fb_x0[k] = f - y * W;
      fb_ysel[k] = 4 * (y < 62 ? y : 62);  // y and y+1 must both be empty rows when clamped
    }
    rowmask = (lane < H) ? (0xFFFFFFFFu >> (32 - W)) : 0u;
  }

  // ---- load / store ---------------------------------------------------------------------
  __device__ __forceinline__ void load_hdr(const uint32_t* hdr_env) {
    const int lane = lane_id();
    uint32_t hv = (lane < HDR_DW) ? hdr_env[lane] : 0u;
    turn = (int)rdlane(hv, H_TURN);
    uint32_t dims = rdlane(hv, H_DIMS);
    W = (int)(dims & 0xFFu);
    H = (int)((dims >> 8) & 0xFFu);
    P = (int)((dims >> 16) & 0xFFu);
    hflags = dims >> 24;
    N = W * H;
    uint32_t st = rdlane(hv, H_STATUS);
    alive = st & 0xFFu;
    winner = (int)((st >> 8) & 0xFFu) - 1;
    last_err = (st >> 16) & 0xFFu;
    episode = (int)rdlane(hv, H_EPISODE);
    recipW = (int)rdlane(hv, H_RECIPW);
    cnt_steps = rdlane(hv, H_CNT_STEPS);
    cnt_abort = rdlane(hv, H_CNT_ABORT);
    cnt_done = rdlane(hv, H_CNT_DONE);
#pragma unroll
    for (int p = 0; p < MAXP; ++p) {
      army_count[p] = (int)rdlane(hv, H_ARMYCNT + p);
      gidx[p] = (int)rdlane(hv, H_GIDX + p);
    }
  }

  __device__ __forceinline__ void store_hdr(uint32_t* hdr_env) const {
    const int lane = lane_id();
    uint32_t v = 0u;
    v = (lane == H_TURN) ? (uint32_t)turn : v;
    v = (lane == H_DIMS) ? ((uint32_t)W | ((uint32_t)H << 8) | ((uint32_t)P << 16) | (hflags << 24)) : v;
    v = (lane == H_STATUS) ? (alive | ((uint32_t)(winner + 1) << 8) | (last_err << 16)) : v;
    v = (lane == H_EPISODE) ? (uint32_t)episode : v;
    v = (lane == H_RECIPW) ? (uint32_t)recipW : v;
    v = (lane == H_CNT_STEPS) ? cnt_steps : v;
    v = (lane == H_CNT_ABORT) ? cnt_abort : v;
    v = (lane == H_CNT_DONE) ? cnt_done : v;
#pragma unroll
    for (int p = 0; p < MAXP; ++p) {
      v = (lane == H_ARMYCNT + p) ? (uint32_t)army_count[p] : v;
      v = (lane == H_GIDX + p) ? (uint32_t)gidx[p] : v;
    }
    if (lane < HDR_DW) hdr_env[lane] = v;
  }

  // rows: HBM --(flat dwordx4)--> LDS --(ds_read_b32, lane = row)--> registers
  __device__ __forceinline__ void load_rows(const uint32_t* rows_env, uint32_t* lds, int hs, int row_dw) {
    const int lane = lane_id();
    for (int c = lane; c * 4 < row_dw; c += 64) {
      uint4 q = *reinterpret_cast<const uint4*>(rows_env + 4 * c);
      *reinterpret_cast<uint4*>(lds + 4 * c) = q;
    }
    wave_lds_fence();
    const bool on = lane < hs;
    const uint32_t* l = lds + (on ? lane : 0);
#pragma unroll
    for (int p = 0; p < MAXP; ++p) {
      uint32_t a = l[(PL::OWN + p) * hs], b = l[(PL::LST + p) * hs], c = l[(PL::VIS + p) * hs];
      own[p] = on ? a : 0u;
      lst[p] = on ? b : 0u;
      vis[p] = on ? c : 0u;
    }
    uint32_t a = l[PL::CHG * hs], b = l[PL::VCH * hs], c = l[PL::GEN * hs], d = l[PL::CITY * hs], e = l[PL::MTN * hs];
    chg = on ? a : 0u;
    vch = on ? b : 0u;
    gen = on ? c : 0u;
    city = on ? d : 0u;
    mtn = on ? e : 0u;
    wave_lds_fence();
  }

  __device__ __forceinline__ void store_rows(uint32_t* rows_env, uint32_t* lds, int hs, bool with_types) const {
    const int lane = lane_id();
    if (lane < hs) {
      uint32_t* l = lds + lane;
#pragma unroll
      for (int p = 0; p < MAXP; ++p) {
        l[(PL::OWN + p) * hs] = own[p];
        l[(PL::LST + p) * hs] = lst[p];
        l[(PL::VIS + p) * hs] = vis[p];
      }
      l[PL::CHG * hs] = chg;
      l[PL::VCH * hs] = vch;
      if (with_types) {
        l[PL::GEN * hs] = gen;
        l[PL::CITY * hs] = city;
        l[PL::MTN * hs] = mtn;
      }
    }
    wave_lds_fence();
    const int ndw = (with_types ? PL::COUNT : PL::MUTABLE) * hs;
    for (int c = lane; c * 4 < ndw; c += 64) {
      // the tail chunk may cover the first dwords of the (unchanged) type planes: those
      // were staged into LDS by load_rows, so the bytes written back are identical.
      uint4 q = *reinterpret_cast<const uint4*>(lds + 4 * c);
      *reinterpret_cast<uint4*>(rows_env + 4 * c) = q;
    }
    wave_lds_fence();
  }

  __device__ __forceinline__ void load_army(const int32_t* army_env) {
    const int lane = lane_id();
#pragma unroll
    for (int s = 0; s < NSLOT; ++s) army[s] = army_env[64 * s + lane];
  }
  __device__ __forceinline__ void store_army(int32_t* army_env) const {
    const int lane = lane_id();
#pragma unroll
    for (int s = 0; s < NSLOT; ++s) army_env[64 * s + lane] = army[s];
  }

  // ---- uniform tile access ----------------------------------------------------------------
  // (readlane of every slot, then a scalar select: a select chain over army[i] itself is
  // rewritten by LLVM into one load through a selected ADDRESS, which demotes the whole
  // board to scratch memory)
  __device__ __forceinline__ int army_get(int t) const {
    const int s = t >> 6, l = t & 63;
    int r = 0;
#pragma unroll
    for (int i = 0; i < NSLOT; ++i) {
      const int v = __builtin_amdgcn_readlane(army[i], l);
      r = (s == i) ? v : r;
    }
    return r;
  }
  __device__ __forceinline__ void army_set(int t, int val) {
    const int s = t >> 6, l = t & 63;
    const bool me = lane_id() == l;
#pragma unroll
    for (int i = 0; i < NSLOT; ++i) army[i] = (me && s == i) ? val : army[i];
  }
  __device__ __forceinline__ bool bit_at(uint32_t plane, int x, int y) const { return (rdlane(plane, y) >> x) & 1u; }

  // tile-domain 0/1 predicate of a row-domain plane
  __device__ __forceinline__ uint32_t gather(uint32_t plane, int s) const { return (bperm(ysel[s], plane) >> xsh[s]) & 1u; }
  // tile-domain all-ones / zero mask of a row-domain plane
  __device__ __forceinline__ int32_t gather_mask(uint32_t plane, int s) const {
    return -(int32_t)((bperm(ysel[s], plane) >> xsh[s]) & 1u);
  }

  // bits [8j, 8j+8) of the plane's row-major bit string (j = lane + 64*pass)
  __device__ __forceinline__ uint32_t flat_byte(uint32_t plane, int pass) const {
    if (W >= 8) {  // 8 consecutive tiles touch at most two rows
      uint32_t r0 = bperm(fb_ysel[pass], plane), r1 = bperm(fb_ysel[pass] + 4, plane);
      uint64_t two = (uint64_t)r0 | ((uint64_t)r1 << W);
      return (uint32_t)(two >> fb_x0[pass]) & 0xFFu;
    }
    uint32_t out = 0u;  // tiny boards (tests): bit by bit
    const int j = lane_id() + 64 * pass;
    for (int b = 0; b < 8; ++b) {
      int f = 8 * j + b;
      int y = (f * recipW) >> 16;
      int x = f - y * W;
      out |= ((bperm(4 * (y < 63 ? y : 63), plane) >> x) & 1u) << b;
    }
    return out;
  }

  // ---- 3x3 / 5x5 dilations (visibility_optimized.go:9-13, :104-105) -----------------------
  __device__ __forceinline__ uint32_t dil3(uint32_t m) const {
    uint32_t h = m | (m << 1) | (m >> 1);
    return (h | from_above(h) | from_below(h)) & rowmask;
  }
  __device__ __forceinline__ uint32_t dil5(uint32_t m) const {
    uint32_t h = m | (m << 1) | (m >> 1) | (m << 2) | (m >> 2);
    uint32_t a = from_above(h), b = from_below(h);
    return (h | a | b | from_above(a) | from_below(b)) & rowmask;
  }

  // ---- Engine.updateFogOfWarOptimized (visibility_optimized.go:16-97) --------------------
  __device__ __forceinline__ void update_fog() {
    if (!(hflags & HF_FOG)) return;  // :17-19
    const int nv = (int)wave_sum(__builtin_popcount(vch));
    if (turn == 0 || nv > N / 10) {  // :22-26 full: clear, then 3x3 around every listed tile of alive players (:33-53)
#pragma unroll
      for (int p = 0; p < MAXP; ++p) vis[p] = ((alive >> p) & 1u) ? dil3(lst[p]) : 0u;
      return;
    }
    if (nv == 0) return;  // incremental update over an empty set is the identity
    // :56-97 affected = board owners within 5x5 of V (:100-116); clear all players in 3x3 of V
    // (:132-150); re-light affected, alive players from their lists (:85-94)
    const uint32_t near5 = dil5(vch), clr = ~dil3(vch);
#pragma unroll
    for (int p = 0; p < MAXP; ++p) {
      const bool affected = wave_any((own[p] & near5) != 0u);
      uint32_t v = vis[p] & clr;
      if (affected && ((alive >> p) & 1u)) v |= dil3(lst[p]);
      vis[p] = v;
    }
  }

  // ---- Engine.updatePlayerStats (stats.go:8-144) -------------------------------------------
  __device__ __forceinline__ void update_stats() {
    const int nc = (int)wave_sum(__builtin_popcount(chg));
    if (nc == 0 && turn > 0) return;             // :10-14
    const bool full = turn == 0 || nc > N / 5;  // :20-21
#pragma unroll
    for (int p = 0; p < MAXP; ++p) lst[p] = full ? own[p] : (own[p] & (lst[p] | chg));  // :33-49 / :90-130
    int32_t acc[MAXP];
#pragma unroll
    for (int p = 0; p < MAXP; ++p) acc[p] = 0;
#pragma unroll
    for (int s = 0; s < NSLOT; ++s) {
#pragma unroll
      for (int p = 0; p < MAXP; ++p) acc[p] += army[s] & gather_mask(lst[p], s);
    }
    alive = 0u;
#pragma unroll
    for (int p = 0; p < MAXP; ++p) {
      army_count[p] = (int)wave_sum((uint32_t)acc[p]);
      // GeneralIdx: the reference keeps the last general in list order (:46,:101,:122); with two
      // or more generals that order depends on Go map iteration.  Here: the highest tile index.
      const uint32_t g = lst[p] & gen;
      const unsigned long long rowsg = __builtin_amdgcn_ballot_w64(g != 0u);
      int gi = -1;
      if (rowsg) {
        const int y = 63 - __builtin_clzll(rowsg);
        const uint32_t r = rdlane(g, y);
        gi = y * W + (31 - __builtin_clz(r));
      }
      gidx[p] = gi;
      alive |= (gi >= 0) ? (1u << p) : 0u;  // :52-54 / :133-135
    }
  }

  // ---- ProductionManager.ProcessTurnProduction (production_manager.go:26-101) -------------
  __device__ __forceinline__ void production(int pg, int pc, int pn, int interval) {
    const bool grow = (turn % interval) == 0;  // :27
    uint32_t listed_alive = 0u;                // :39-45: lists of alive players, owner NOT re-checked (H7)
#pragma unroll
    for (int p = 0; p < MAXP; ++p) listed_alive |= ((alive >> p) & 1u) ? lst[p] : 0u;
    const uint32_t normal = ~(gen | city | mtn) & rowmask;
    const uint32_t mg = (pg > 0) ? (listed_alive & gen) : 0u;
    const uint32_t mc = (pc > 0) ? (listed_alive & city) : 0u;
    const uint32_t mn = (grow && pn > 0) ? (listed_alive & normal) : 0u;
    chg |= mg | mc | mn;  // :59-61 (prod > 0 only)
    const int an = (grow && pn > 0) ? pn : pc;
    if (pg == pc && pc == an) {  // one rate for every producing tile: one gather per slot
      const uint32_t m = mg | mc | mn;
#pragma unroll
      for (int s = 0; s < NSLOT; ++s) army[s] += pg & gather_mask(m, s);
    } else {
#pragma unroll
      for (int s = 0; s < NSLOT; ++s)
        army[s] += (pg & gather_mask(mg, s)) + (pc & gather_mask(mc, s)) + (pn & gather_mask(mn, s));
    }
  }

  // ---- WinConditionChecker.CheckGameOver (rules/win_conditions.go:21-57) -------------------
  __device__ __forceinline__ void check_game_over() {
    const int na = __builtin_popcount(alive);
    const bool over = (P > 1) ? (na <= 1) : (na == 0);  // originalPlayers == len(Players)
    hflags = over ? (hflags | HF_DONE) : (hflags & ~HF_DONE);
    winner = (over && na == 1) ? (31 - __builtin_clz(alive)) : -1;
  }

  // ---- one player's move: MoveAction.Validate + core.ApplyMoveAction ------------------------
  // (core/action.go:56-105, core/movement.go:23-89) inside ActionProcessor.ProcessActions
  // (processor/action_processor.go:36-99).  `orders` packs core.ProcessCaptures' output
  // (movement.go:100-118): byte k = victim | new_owner << 4.
  template <int PID>
  __device__ __forceinline__ void apply_action(uint32_t a_lo, uint32_t a_hi, uint32_t& first_err, uint64_t& orders,
                                               int& n_orders, uint32_t& elim_seen, uint32_t& ncap) {
    if (!(a_hi & GVEC_ACT_VALID)) return;  // nil action
    if (!((alive >> PID) & 1u)) return;     // action_processor.go:56-60 (Alive as last written, H2)
    const int fx = (int)(int8_t)(a_lo & 0xFFu), fy = (int)(int8_t)((a_lo >> 8) & 0xFFu);
    const int tx = (int)(int8_t)((a_lo >> 16) & 0xFFu), ty = (int)(int8_t)(a_lo >> 24);
    uint32_t code = 0u;
    int fa = 0;
    const bool inb_f = fx >= 0 && fx < W && fy >= 0 && fy < H, inb_t = tx >= 0 && tx < W && ty >= 0 && ty < H;
    if (!inb_f || !inb_t) code = GVEC_ERR_INVALID_COORDINATES;       // action.go:58-64
    else if (fx == tx && fy == ty) code = GVEC_ERR_MOVE_TO_SELF;    // :67-69
    else {
      const int dx = fx - tx, dy = fy - ty;
      const bool adj = (dx == 0 && (dy == 1 || dy == -1)) || (dy == 0 && (dx == 1 || dx == -1));
      if (!adj) code = GVEC_ERR_NOT_ADJACENT;  // :72-76
      else if (!bit_at(own[PID], fx, fy)) code = GVEC_ERR_NOT_OWNED;  // :82-84
      else {
        fa = army_get(fy * W + fx);
        if (fa <= 1) code = GVEC_ERR_INSUFFICIENT_ARMY;                // :87-89
        else if (bit_at(mtn, tx, ty)) code = GVEC_ERR_TARGET_IS_MOUNTAIN;  // :96-98
      }
    }
    if (code) {  // action_processor.go:66-77: remember the FIRST error, keep going
      first_err = first_err ? first_err : code;
      return;
    }
    int n = (a_hi & GVEC_ACT_HALF) ? (fa / 2) : (fa - 1);  // movement.go:40-49
    n = (n == 0) ? 1 : n;
    const int lane = lane_id();
    const uint32_t fbit = (lane == fy) ? (1u << fx) : 0u, tbit = (lane == ty) ? (1u << tx) : 0u;
    army_set(fy * W + fx, fa - n);  // :54
    chg |= fbit | tbit;             // :57-60
    const int tt = ty * W + tx;
    const int ta = army_get(tt);
    if (bit_at(own[PID], tx, ty)) {  // :62-66 own tile: consolidate
      army_set(tt, ta + n);
    } else if (n > ta) {  // :69-82 capture (ties favour the defender)
      int prev = -1;
#pragma unroll
      for (int q = 0; q < MAXP; ++q) {
        if (bit_at(own[q], tx, ty)) prev = q;
        own[q] &= ~tbit;
      }
      own[PID] |= tbit;
      army_set(tt, n - ta);
      vch |= tbit;  // action_processor.go:84-86
      ncap++;
      // movement.go:105-108
      if (bit_at(gen, tx, ty) && prev >= 0 && !((elim_seen >> prev) & 1u)) {
        orders |= (uint64_t)((uint32_t)prev | ((uint32_t)PID << 4)) << (8 * n_orders);
        n_orders++;
        elim_seen |= 1u << prev;
      }
    } else {
      army_set(tt, ta - n);  // :85
    }
  }

  template <int PID>
  __device__ __forceinline__ void act_chain(uint32_t acts_lo, uint32_t acts_hi, uint32_t& first_err, uint64_t& orders,
                                            int& n_orders, uint32_t& elim_seen, uint32_t& ncap) {
    if constexpr (PID < MAXP) {
      if (PID < P)
        apply_action<PID>(rdlane(acts_lo, PID), rdlane(acts_hi, PID), first_err, orders, n_orders, elim_seen, ncap);
      act_chain<PID + 1>(acts_lo, acts_hi, first_err, orders, n_orders, elim_seen, ncap);
    }
  }

  // ---- Engine.handleEliminationsAndTileTurnover (engine.go:118-152) -------------------------
  __device__ __forceinline__ void eliminate(uint64_t orders, int n_orders) {
    for (int k = 0; k < n_orders; ++k) {
      const int v = (int)((orders >> (8 * k)) & 15u), nw = (int)((orders >> (8 * k + 4)) & 15u);
      uint32_t tiles = 0u;  // victim's listed tiles still owned by the victim (:130-131, H4)
#pragma unroll
      for (int q = 0; q < MAXP; ++q) tiles |= (q == v) ? (lst[q] & own[q]) : 0u;
#pragma unroll
      for (int q = 0; q < MAXP; ++q) {
        own[q] = (q == v) ? (own[q] & ~tiles) : own[q];
        own[q] = (q == nw) ? (own[q] | tiles) : own[q];
        gidx[q] = (q == v) ? -1 : gidx[q];  // :141
      }
      chg |= tiles;  // :133-134
      vch |= tiles;
      alive &= ~(1u << v);  // :140
    }
  }

  // ---- TurnProcessor.ProcessTurn (turn_processor.go:29-77) ----------------------------------
  // acts_lo/hi: lane p holds player p's gvec_action words.  Returns the per-env error code.
  __device__ __forceinline__ uint32_t turn_step(uint32_t acts_lo, uint32_t acts_hi, const StepArgs& A, uint32_t& ncap_out,
                                                bool& aborted) {
    aborted = false;
    ncap_out = 0u;
    if (hflags & HF_DONE) return GVEC_ERR_GAME_OVER;  // validateGameState :95-113
    turn++;                                           // initializeTurn :124-135
    update_fog();
    chg = 0u;
    vch = 0u;
    uint32_t first_err = 0u, elim_seen = 0u, ncap = 0u;
    uint64_t orders = 0ull;
    int n_orders = 0;
    // Engine.processActions (engine.go:80-115): PlayerID order == slot order (sort.Slice :39-41)
    act_chain<0>(acts_lo, acts_hi, first_err, orders, n_orders, elim_seen, ncap);
    ncap_out = ncap;
    if (n_orders > 0) {  // engine.go:101-109
      eliminate(orders, n_orders);
      update_stats();
    }
    if (first_err) {  // engine.go:111-113 -> turn_processor.go:55-57: production, stats, game-over skipped (H5)
      aborted = true;
      return first_err;
    }
    production(A.prod_general, A.prod_city, A.prod_normal, A.interval);  // :60
    update_stats();                                                       // :65,170-179
    check_game_over();
    return 0u;
  }

  // ---- EngineInitializer.performInitialSetup (engine_initializer.go:218-225) -----------------
  __device__ __forceinline__ void initial_setup() {
    turn = 0;
    chg = 0u;
    vch = 0u;
#pragma unroll
    for (int p = 0; p < MAXP; ++p) vis[p] = 0u;
    hflags &= ~HF_DONE;
    update_stats();  // Turn == 0 => full
    update_fog();    // Turn == 0 => full
    check_game_over();
  }

  // ---- LegalMoveCalculator.GetLegalActionMask (rules/legal_moves.go:19-73) --------------------
  // out[p][k]: lane j holds bits [32*(j+64k), +32) of player p's mask, bit i = action
  // (y*W+x)*4+d with d = 0 up, 1 right, 2 down, 3 left (H10).
  __device__ __forceinline__ void legal_masks(uint32_t (&out)[MAXP][MPASS]) const {
    const int lane = lane_id();
    // army > 1 as a flat bit string: dword i in lane i
    uint32_t gt1_dw = 0u;
#pragma unroll
    for (int s = 0; s < NSLOT; ++s) {
      const unsigned long long b = __builtin_amdgcn_ballot_w64(army[s] > 1);
      gt1_dw = (lane == 2 * s) ? (uint32_t)b : gt1_dw;
      gt1_dw = (lane == 2 * s + 1) ? (uint32_t)(b >> 32) : gt1_dw;
    }
    const uint32_t notm = ~mtn & rowmask;  // in-board, not a mountain (Validate :58-64,:96-98)
    const uint32_t ok_up = from_above(notm) & rowmask, ok_dn = from_below(notm) & rowmask;
    const uint32_t ok_rt = notm >> 1, ok_lf = (notm << 1) & rowmask;
#pragma unroll
    for (int k = 0; k < MPASS; ++k) {
      const int j = lane + 64 * k;
      const uint32_t okn = spread4(flat_byte(ok_up, k)) | (spread4(flat_byte(ok_rt, k)) << 1) |
                           (spread4(flat_byte(ok_dn, k)) << 2) | (spread4(flat_byte(ok_lf, k)) << 3);
      const uint32_t gt1 = (bperm((j >> 2) * 4, gt1_dw) >> (8 * (j & 3))) & 0xFFu;
#pragma unroll
      for (int p = 0; p < MAXP; ++p) {
        // :26-28 alive, :37 listed, :41 owner == pid && army > 1
        const uint32_t cb = flat_byte(lst[p] & own[p], k) & gt1;  // cross-lane: keep it unconditional
        const uint32_t can = ((alive >> p) & 1u) ? cb : 0u;
        out[p][k] = (spread4(can) * 15u) & okn;
      }
    }
  }
};

}  // namespace gvec
