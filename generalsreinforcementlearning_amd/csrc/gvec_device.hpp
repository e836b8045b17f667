// gvec_device.hpp — CDNA4 (gfx950) device code of the batched Generals.io turn engine.
//
// One 64-lane wavefront owns one board ("env").  The board lives in registers in two
// layouts at once (DESIGN.md "Data layout"):
//
//   * FLAT domain — every bit-plane is the row-major bit string of the board (bit t = tile
//     t = y*W + x, the reference's index, core/board.go:108); lane i holds bits 32i..32i+31:
//       own[p]  tile.Owner == p                  (core/board.go:8)
//       lst[p]  tile in Players[p].OwnedTiles    (game/state.go:12; SURVEY H6 "plane L")
//       vis[p]  bit p of tile.VisibleBitfield    (core/board.go:11)
//       chg     GameState.ChangedTiles, vch GameState.VisibilityChangedTiles (state.go:29,33)
//       gen/city/mtn  tile.Type one-hot          (core/board.go:20-26)
//     Neighbours are funnel shifts of the string by 1 (x +- 1, with column guards) and by W
//     (y +- 1): a DPP wave shift + v_alignbit each.  Ownership algebra and the legal-move
//     predicate are plain ANDs.  13 lanes carry a 20x20 board, 32 lanes a 32x32 one.
//   * TILE domain — lane l, slot s holds Tile.Army of tile t = 64*s + l (int32, H12).
//     A flat plane reaches the tile domain with one ds_bpermute (dword 2s + (l>>5)) and a
//     bit-field extract at bit l&31: no per-tile coordinates are ever computed.
//
// Planes and armies are loaded straight into registers (plane p: lane i reads dword i; armies:
// 256 B per wave instruction); the planes block of one env is a contiguous, 16-byte aligned run.  Integer / bit work only: no MFMA
// anywhere (HBM-roofline kernel).
//
// Every routine cites the Go function it reproduces (paths relative to
// /root/reference/internal/game/).  Semantics are the plane re-statement derived in
// SURVEY.md section 8 (H1-H12); tests/ checks them bit-for-bit against oracle/.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/generals_vec.h"

// clang has no __builtin for v_writelane_b32: bind the LLVM intrinsic directly (value and lane select must be
// wave-uniform; the backend moves a run-time lane select to M0)
extern "C" __device__ int gvec_llvm_writelane(int value, int lane, int vdst_in) __asm("llvm.amdgcn.writelane.i32");

namespace gvec {

// ---- resident record layout (per env) -------------------------------------------------
// hdr  : HDR_DW u32      planes : M*FD u32 (+pad to 4), plane-major [m][i]
// army : NARROW block, NSLOT*64 u16 (the default), or - header flag HF_WIDE - the env's block of the
//        int32 escape array, NSLOT*64 i32 (see "army storage" below)
constexpr int HDR_DW = 24;
enum : int {
  H_TURN = 0,     // GameState.Turn
  H_DIMS = 1,     // W | H<<8 | P<<16 | flags<<24   (flags: bit0 Engine.gameOver, bit1 FogOfWarEnabled, bit2 wide armies)
  H_STATUS = 2,   // alive bits | last err<<16
  H_EPISODE = 3,  // re-deal counter (auto-reset)
  H_ARMYCNT = 4,  // [8] Player.ArmyCount
  H_GIDX = 12,    // [8] Player.GeneralIdx
  H_RECIPW = 20,  // ceil(65536 / W): exact t / W for t < 1024
  H_CNT_STEPS = 21,  // lifetime counters of this env slot (rollout statistics)
  H_CNT_ABORT = 22,
  H_CNT_DONE = 23
};
constexpr uint32_t HF_DONE = 1u, HF_FOG = 2u, HF_WIDE = 4u;

// plane order inside the planes block; the last three never change after reset
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
  uint32_t* rows;    // the planes block
  uint32_t* army16;  // narrow armies, NSLOT*32 dwords per env
  int32_t* army32;   // wide escape, NSLOT*64 dwords per env
  const gvec_action* actions;  // [B][pstride] (ignored with KF_AGENT)
  gvec_action* actions_out;    // optional: where the agent records what it played
  int32_t* err;                // [B] or null
  uint32_t* legal;             // [B][pstride][mask_dw]
  const uint32_t* pool_hdr;
  const uint32_t* pool_rows;
  const uint32_t* pool_army16;
  const int32_t* pool_army32;
  int32_t num_envs, fd, row_dw, mask_dw, pool_size;  // fd = dwords per plane = ceil(max tiles / 32)
  int32_t pstride;  // players per env in actions / legal buffers (gvec_config.max_players)
  int32_t prod_general, prod_city, prod_normal, interval;
  int32_t turns, invalid_permille;
  uint32_t agent_noop, agent_half;  // gvec_set_agent_mix thresholds (of 65536)
  uint32_t flags, seed_lo, seed_hi, pool_seed_lo, pool_seed_hi;
};

// ---- army storage ------------------------------------------------------------------------
// Tile.Army is a Go int (core/board.go:9).  On the device it is computed in int32 registers and
// stored in one of two forms, chosen per env at every store:
//   NARROW  u16, when every army of the board is in [0, 65535] (virtually always): two 64-tile slots
//           share a dword - dword 64k+l = army(slot 2k, lane l) | army(slot 2k+1, lane l) << 16 - so one
//           256-byte wave access moves 128 tiles; an odd last slot is 64 halfwords (one 128-byte line).
//           NSLOT*128 bytes per env: 896 B instead of 1,792 B at 20x20.
//   WIDE    int32 in the env's block of the escape array (header flag HF_WIDE), as soon as one army
//           leaves that range.  Exact for everything an int32 can hold; the env returns to NARROW at
//           the first store where it fits again.
// Nothing is ever clamped: the pair (flag, block) always holds the exact int32 value.
struct ArmyRef {
  uint32_t* n;  // this env's narrow block (NSLOT*32 dwords)
  int32_t* w;   // this env's wide block   (NSLOT*64 dwords)
};
struct ArmyCRef {
  const uint32_t* n;
  const int32_t* w;
  __device__ __forceinline__ ArmyCRef(const uint32_t* n_, const int32_t* w_) : n(n_), w(w_) {}
  __device__ __forceinline__ ArmyCRef(const ArmyRef& r) : n(r.n), w(r.w) {}
};
template <int NSLOT>
__device__ __forceinline__ ArmyRef army_ref(uint32_t* a16, int32_t* a32, int env) {
  return ArmyRef{a16 + (size_t)env * (NSLOT * 32), a32 + (size_t)env * (NSLOT * 64)};
}
template <int NSLOT>
__device__ __forceinline__ ArmyCRef army_cref(const uint32_t* a16, const int32_t* a32, int env) {
  return ArmyCRef(a16 + (size_t)env * (NSLOT * 32), a32 + (size_t)env * (NSLOT * 64));
}
template <int NSLOT>
__device__ __forceinline__ void army_load_narrow(int32_t (&army)[NSLOT], const uint32_t* n) {
  const int lane = (int)(threadIdx.x & 63u);
#pragma unroll
  for (int k = 0; k < NSLOT / 2; ++k) {
    const uint32_t w = n[64 * k + lane];
    army[2 * k] = (int32_t)(w & 0xFFFFu);
    army[2 * k + 1] = (int32_t)(w >> 16);
  }
  if constexpr ((NSLOT & 1) != 0) army[NSLOT - 1] = (int32_t)reinterpret_cast<const uint16_t*>(n + 64 * (NSLOT / 2))[lane];
}
template <int NSLOT>
__device__ __forceinline__ void army_load_wide(int32_t (&army)[NSLOT], const int32_t* w) {
  const int lane = (int)(threadIdx.x & 63u);
#pragma unroll
  for (int s = 0; s < NSLOT; ++s) army[s] = w[64 * s + lane];
}
// wave-uniform: every army of the board fits the narrow form (negative values have bit 31 set)
template <int NSLOT>
__device__ __forceinline__ bool army_fits_narrow(const int32_t (&army)[NSLOT]) {
  uint32_t m = 0u;
#pragma unroll
  for (int s = 0; s < NSLOT; ++s) m |= (uint32_t)army[s];
  return __builtin_amdgcn_ballot_w64(m > 0xFFFFu) == 0ull;
}
template <int NSLOT>
__device__ __forceinline__ void army_store_narrow(const int32_t (&army)[NSLOT], uint32_t* n) {
  const int lane = (int)(threadIdx.x & 63u);
#pragma unroll
  for (int k = 0; k < NSLOT / 2; ++k) n[64 * k + lane] = (uint32_t)army[2 * k] | ((uint32_t)army[2 * k + 1] << 16);
  if constexpr ((NSLOT & 1) != 0) reinterpret_cast<uint16_t*>(n + 64 * (NSLOT / 2))[lane] = (uint16_t)army[NSLOT - 1];
}
template <int NSLOT>
__device__ __forceinline__ void army_store_wide(const int32_t (&army)[NSLOT], int32_t* w) {
  const int lane = (int)(threadIdx.x & 63u);
#pragma unroll
  for (int s = 0; s < NSLOT; ++s) w[64 * s + lane] = army[s];
}

// ---- wave primitives --------------------------------------------------------------------
// NOTE: ds_bpermute / DPP read 0 from lanes that are masked off in EXEC.  Every cross-lane
// helper below must therefore be called under wave-uniform control flow only.
__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63u); }

template <int CTRL, int RM = 0xf, int BM = 0xf>
__device__ __forceinline__ uint32_t dpp0(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, RM, BM, false);
}
// lane i receives lane i-1 (the 32 tiles before); lane 0 receives 0     [DPP wave_shr:1]
__device__ __forceinline__ uint32_t from_prev(uint32_t v) { return dpp0<0x138>(v); }
// lane i receives lane i+1 (the 32 tiles after); lane 63 receives 0     [DPP wave_shl:1]
__device__ __forceinline__ uint32_t from_next(uint32_t v) { return dpp0<0x130>(v); }

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
// a * b + c on the low 24 bits of a and b, full rate.  Written as one instruction because the compiler
// otherwise splits it into v_mul_u32_u24 + a shared v_add3 (2.5 instead of 2 instructions per term).
__device__ __forceinline__ uint32_t mad24(uint32_t a, uint32_t b, uint32_t c) {
  uint32_t d;
  asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
__device__ __forceinline__ uint32_t rdlane(uint32_t v, int lane) {
  return (uint32_t)__builtin_amdgcn_readlane((int)v, lane);
}
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ uint32_t wave_sum(uint32_t v) { return rdlane(wave_scan_add(v), 63); }
__device__ __forceinline__ bool wave_any(bool c) { return __builtin_amdgcn_ballot_w64(c) != 0ull; }
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

// kColumnPattern[W] = sum of 1 << k for k = 0, W, 2W, ... < 32: where column 0 falls in a 32-tile
// window that starts in column 0
struct ColumnPatternTable {
  uint32_t v[33];
  constexpr ColumnPatternTable() : v{} {
    for (int w = 1; w <= 32; ++w) {
      uint32_t p = 0u;
      for (int k = 0; k < 32; k += w) p |= 1u << k;
      v[w] = p;
    }
  }
  __device__ __forceinline__ uint32_t operator[](int w) const { return v[w]; }
};
__device__ const ColumnPatternTable kColumnPattern{};

// =========================================================================================
template <int MAXP, int NSLOT>
struct Board {
  using PL = Planes<MAXP>;
  static constexpr int MPASS = (NSLOT > 8) ? 2 : 1;  // legal-mask dwords per lane (8 tiles each)

  // flat domain
  uint32_t own[MAXP], lst[MAXP], vis[MAXP], chg, vch, gen, city, mtn;
  uint32_t valid;  // bits t < N
  uint32_t ncol0;  // bits with x != 0      (guards a shift towards higher t)
  uint32_t ncolL;  // bits with x != W - 1  (guards a shift towards lower t)
  // tile domain
  int32_t army[NSLOT];
  // header: lane k holds header dword k (Player.ArmyCount / GeneralIdx, counters, ... live here;
  // only what the turn logic branches on is also kept wave-uniform below)
  uint32_t hv;
  int32_t* larmy;  // LDS shadow of the armies during the action phase: tile t at larmy[t]
  // wave-uniform
  int W, H, P, N, turn, recipW;
  uint32_t alive, hflags;
  // every army stays below 2^23 until the board is stored (true for one turn from a NARROW load): the
  // masked sums may then use full-rate 24-bit multiply-adds.  Set by the kernel, never by load_*.
  bool small = false;

  // ---- geometry masks of this board size ----------------------------------------------------
  __device__ __forceinline__ void geometry() {
    const int t0 = 32 * (lane_id() & 31);
    const int left = N - t0;
    valid = (lane_id() >= 32 || left <= 0) ? 0u : (left >= 32 ? 0xFFFFFFFFu : ((1u << left) - 1u));
    const uint32_t pat = kColumnPattern[W];  // bits at multiples of W below 32 (one scalar load)
    // 24-bit multiplies are full rate (v_mul_lo_u32 is quarter rate); t0 < 2048, recipW <= 65536, W <= 32
    const int q = (int)(__umul24((uint32_t)t0, (uint32_t)recipW) >> 16);  // t0 / W, exact for t0 < 1024
    const int x0 = t0 - (int)__umul24((uint32_t)q, (uint32_t)W);          // column of this lane's first tile
    const uint32_t col0 = pat << (x0 ? W - x0 : 0);
    ncol0 = ~col0;
    ncolL = ~__builtin_amdgcn_alignbit(from_next(col0), col0, 1);  // t is in the last column iff t+1 is in column 0
  }

  // ---- flat-string shifts: bit t of the result = bit (t -+ k) of m ---------------------------
  __device__ __forceinline__ uint32_t up1(uint32_t m) const { return __builtin_amdgcn_alignbit(m, from_prev(m), 31); }  // from t-1
  __device__ __forceinline__ uint32_t dn1(uint32_t m) const { return __builtin_amdgcn_alignbit(from_next(m), m, 1); }   // from t+1
  __device__ __forceinline__ uint32_t upW(uint32_t m) const {                                                           // from t-W
    return __builtin_amdgcn_alignbit(m, from_prev(m), (uint32_t)(32 - W) & 31u);
  }
  __device__ __forceinline__ uint32_t dnW(uint32_t m) const {  // from t+W
    return (uint32_t)((((uint64_t)from_next(m) << 32) | (uint64_t)m) >> W);
  }
  __device__ __forceinline__ uint32_t dil_h(uint32_t m) const { return m | (up1(m) & ncol0) | (dn1(m) & ncolL); }
  __device__ __forceinline__ uint32_t dil_v(uint32_t m) const { return (m | upW(m) | dnW(m)) & valid; }
  // 3x3 / 5x5 neighbourhoods (visibility_optimized.go:9-13, :104-105)
  __device__ __forceinline__ uint32_t dil3(uint32_t m) const { return dil_v(dil_h(m)); }

  // ---- load / store ---------------------------------------------------------------------
  __device__ __forceinline__ void load_hdr(const uint32_t* hdr_env) {
    const int lane = lane_id();
    hv = (lane < HDR_DW) ? hdr_env[lane] : 0u;
    turn = (int)rdlane(hv, H_TURN);
    const uint32_t dims = rdlane(hv, H_DIMS);
    W = (int)(dims & 0xFFu);
    H = (int)((dims >> 8) & 0xFFu);
    P = (int)((dims >> 16) & 0xFFu);
    hflags = dims >> 24;
    N = W * H;
    alive = rdlane(hv, H_STATUS) & 0xFFu;
    recipW = (int)rdlane(hv, H_RECIPW);
  }
  // v must be wave-uniform (every caller passes scalar values): one v_writelane, no compare / select
  __device__ __forceinline__ void hdr_set(int k, uint32_t v) { hv = (uint32_t)gvec_llvm_writelane((int)v, k, (int)hv); }
  __device__ __forceinline__ uint32_t hdr_get(int k) const { return rdlane(hv, k); }

  __device__ __forceinline__ void store_hdr(uint32_t* hdr_env, uint32_t last_err) {
    const int lane = lane_id();
    hdr_set(H_TURN, (uint32_t)turn);
    hdr_set(H_DIMS, (uint32_t)W | ((uint32_t)H << 8) | ((uint32_t)P << 16) | (hflags << 24));
    hdr_set(H_STATUS, alive | (last_err << 16));
    hdr_set(H_RECIPW, (uint32_t)recipW);
    if (lane < HDR_DW) hdr_env[lane] = hv;
  }

  // planes: lane i holds dword i of each bit string.  Staging the block through LDS with dwordx4
  // copies was measured 4 % slower (the LDS pipe is as busy as the VALU in this kernel).
  __device__ __forceinline__ void load_planes(const uint32_t* rows_env, int fd) {
    const int lane = lane_id();
    const bool on = lane < fd;
    const uint32_t* g = rows_env + (on ? lane : 0);
#pragma unroll
    for (int p = 0; p < MAXP; ++p) {
      uint32_t a = g[(PL::OWN + p) * fd], b = g[(PL::LST + p) * fd], c = g[(PL::VIS + p) * fd];
      own[p] = on ? a : 0u;
      lst[p] = on ? b : 0u;
      vis[p] = on ? c : 0u;
    }
    uint32_t a = g[PL::CHG * fd], b = g[PL::VCH * fd], c = g[PL::GEN * fd], d = g[PL::CITY * fd], e = g[PL::MTN * fd];
    chg = on ? a : 0u;
    vch = on ? b : 0u;
    gen = on ? c : 0u;
    city = on ? d : 0u;
    mtn = on ? e : 0u;
  }

  // The type planes change only when the env is re-dealt (with_types).
  __device__ __forceinline__ void store_planes(uint32_t* rows_env, int fd, int row_dw, bool with_types) const {
    const int lane = lane_id();
    if (lane < fd) {
      uint32_t* g = rows_env + lane;
#pragma unroll
      for (int p = 0; p < MAXP; ++p) {
        g[(PL::OWN + p) * fd] = own[p];
        g[(PL::LST + p) * fd] = lst[p];
        g[(PL::VIS + p) * fd] = vis[p];
      }
      g[PL::CHG * fd] = chg;
      g[PL::VCH * fd] = vch;
      if (with_types) {
        g[PL::GEN * fd] = gen;
        g[PL::CITY * fd] = city;
      }
    }
    if (with_types && lane < row_dw - PL::MTN * fd) rows_env[PL::MTN * fd + lane] = mtn;  // lanes >= fd hold 0: the padding
  }

  // whole 64-tile slots travel both ways (the padding beyond N holds zeros).  Trimming the store to
  // the board's N tiles was measured 4 % SLOWER: it turns the last slot into partial-line writes.
  // The narrow block is read unconditionally (its loads need not wait for the header); a WIDE env
  // - rare - reads its escape block on top.  Needs hflags: call after load_hdr.
  __device__ __forceinline__ void load_army(const ArmyCRef& a) {
    army_load_narrow<NSLOT>(army, a.n);
    if (hflags & HF_WIDE) army_load_wide<NSLOT>(army, a.w);
  }
  // Chooses the form (and sets / clears HF_WIDE in hflags accordingly): call BEFORE store_hdr.
  __device__ __forceinline__ void store_army(const ArmyRef& a) {
    if (army_fits_narrow<NSLOT>(army)) {
      hflags &= ~HF_WIDE;
      army_store_narrow<NSLOT>(army, a.n);
    } else {
      hflags |= HF_WIDE;
      army_store_wide<NSLOT>(army, a.w);
    }
  }

  // ---- uniform tile access ----------------------------------------------------------------
  // During the action phase the armies live in an LDS shadow: a wave-uniform tile index is one
  // broadcast ds_read / one single-lane ds_write instead of an NSLOT-way register select chain
  // (which costs ~3 scalar instructions per slot on the CU's single scalar pipe).
  __device__ __forceinline__ void army_to_lds() {
    const int lane = lane_id();
#pragma unroll
    for (int s = 0; s < NSLOT; ++s) larmy[64 * s + lane] = army[s];
    wave_lds_fence();
  }
  __device__ __forceinline__ void army_from_lds() {
    const int lane = lane_id();
    wave_lds_fence();
#pragma unroll
    for (int s = 0; s < NSLOT; ++s) army[s] = larmy[64 * s + lane];
  }
  __device__ __forceinline__ int army_get(int t) const { return uni(larmy[t]); }
  __device__ __forceinline__ void army_set(int t, int val) {
    if (lane_id() == 0) larmy[t] = val;
  }
  __device__ __forceinline__ bool bit_at(uint32_t plane, int t) const { return (rdlane(plane, t >> 5) >> (t & 31)) & 1u; }
  // the single bit of tile t as a flat plane
  __device__ __forceinline__ uint32_t tile_bit(int t) const { return (lane_id() == (t >> 5)) ? (1u << (t & 31)) : 0u; }

  // tile-domain all-ones / zero mask (gather_mask) or 0/1 (gather) of a flat plane: tile 64s+l is
  // bit l&31 of dword 2s + (l>>5)
  __device__ __forceinline__ int32_t gather_mask(uint32_t plane, int s) const {
    const int lane = lane_id();
    return __builtin_amdgcn_sbfe((int32_t)bperm(((lane >> 5) << 2) + 8 * s, plane), (uint32_t)(lane & 31), 1u);
  }
  __device__ __forceinline__ uint32_t gather(uint32_t plane, int s) const {
    const int lane = lane_id();
    return __builtin_amdgcn_ubfe(bperm(((lane >> 5) << 2) + 8 * s, plane), (uint32_t)(lane & 31), 1u);
  }

  // tile-domain predicates -> flat plane (the ballot of slot s is dwords 2s, 2s+1 of the bit string)
  __device__ __forceinline__ void scatter(uint32_t& plane, unsigned long long ballot, int s) const {
    const int lane = lane_id();
    plane = (lane == 2 * s) ? (uint32_t)ballot : plane;
    plane = (lane == 2 * s + 1) ? (uint32_t)(ballot >> 32) : plane;
  }

  // ---- Engine.updateFogOfWarOptimized (visibility_optimized.go:16-97) --------------------
  __device__ __forceinline__ void update_fog() {
    if (!(hflags & HF_FOG)) return;  // :17-19
    const int nv = (int)wave_sum(__builtin_popcount(vch));
    if (turn == 0 || nv > N / 10) {  // :22-26 full: clear, then 3x3 around every listed tile of alive players (:33-53)
#pragma unroll
      for (int p = 0; p < MAXP; ++p) {
        const uint32_t d = dil3(lst[p]);  // cross-lane: outside the (uniform) select
        vis[p] = ((alive >> p) & 1u) ? d : 0u;
      }
      return;
    }
    if (nv == 0) return;  // incremental update over an empty set is the identity
    // :56-97 affected = board owners within 5x5 of V (:100-116); clear all players in 3x3 of V
    // (:132-150); re-light affected, alive players from their lists (:85-94)
    const uint32_t near3 = dil3(vch), near5 = dil3(near3), clr = ~near3;
#pragma unroll
    for (int p = 0; p < MAXP; ++p) {
      const bool relight = wave_any((own[p] & near5) != 0u) && ((alive >> p) & 1u);
      uint32_t v = vis[p] & clr;
      if (relight) v |= dil3(lst[p]);  // wave-uniform branch
      vis[p] = v;
    }
  }

  // Wave-wide sums of MAXP per-lane accumulators in ONE reduction tree (a tree per player costs 6 DPP
  // adds each).  First the accumulators are folded into one register, lane l keeping class
  // l & (MAXP-1): at each level a lane keeps its own class's half and hands the other half to its
  // partner (quad_perm / row_ror swaps).  Then lanes of equal class are summed: row_shr inside a row of
  // 16, ds_swizzle (xor 16) and ds_bpermute (xor 32) across rows.  Returns a register whose lanes
  // H_ARMYCNT+p hold the total of acc[p].
  __device__ __forceinline__ uint32_t multi_sum(const int32_t (&acc)[MAXP]) const {
    static_assert(MAXP == 2 || MAXP == 4 || MAXP == 8, "fold levels are written for 2, 4 or 8 players");
    static_assert(H_ARMYCNT == 4, "the final row_shl assumes header lanes 4..4+MAXP-1");
    const int lane = lane_id();
    uint32_t m[MAXP];
#pragma unroll
    for (int p = 0; p < MAXP; ++p) m[p] = (uint32_t)acc[p];
    {
      const bool hi = (lane & 1) != 0;
#pragma unroll
      for (int i = 0; i < MAXP / 2; ++i) {
        const uint32_t keep = hi ? m[2 * i + 1] : m[2 * i], give = hi ? m[2 * i] : m[2 * i + 1];
        m[i] = keep + dpp0<0xB1>(give);  // quad_perm [1,0,3,2]
      }
    }
    if constexpr (MAXP >= 4) {
      const bool hi = (lane & 2) != 0;
#pragma unroll
      for (int i = 0; i < MAXP / 4; ++i) {
        const uint32_t keep = hi ? m[2 * i + 1] : m[2 * i], give = hi ? m[2 * i] : m[2 * i + 1];
        m[i] = keep + dpp0<0x4E>(give);  // quad_perm [2,3,0,1]
      }
    }
    if constexpr (MAXP >= 8) {
      const bool hi = (lane & 4) != 0;
      const uint32_t keep = hi ? m[1] : m[0], give = hi ? m[0] : m[1];
      m[0] = keep + dpp0<0x124>(give);  // row_ror:4 (the partner differs in bit 2; a bijection is all a sum needs)
    }
    uint32_t r = m[0];
    if constexpr (MAXP <= 2) r += dpp0<0x112>(r);  // row_shr:2
    if constexpr (MAXP <= 4) r += dpp0<0x114>(r);  // row_shr:4
    r += dpp0<0x118>(r);                           // row_shr:8 -> the last MAXP lanes of each row hold the row's sums
    r += (uint32_t)__builtin_amdgcn_ds_swizzle((int)r, 0x401F);  // lane ^ 16
    r += bperm((lane ^ 32) << 2, r);
    return dpp0<0x100 + 12 - MAXP>(r);  // row_shl: lanes 16-MAXP.. of row 0 -> lanes 4..
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
    if (small) {  // bit * army + acc in one full-rate v_mad_u32_u24
#pragma unroll
      for (int s = 0; s < NSLOT; ++s) {
#pragma unroll
        for (int p = 0; p < MAXP; ++p) acc[p] = (int32_t)mad24((uint32_t)army[s], gather(lst[p], s), (uint32_t)acc[p]);
      }
    } else {
#pragma unroll
      for (int s = 0; s < NSLOT; ++s) {
#pragma unroll
        for (int p = 0; p < MAXP; ++p) acc[p] += army[s] & gather_mask(lst[p], s);
      }
    }
    {  // Player.ArmyCount: header lanes H_ARMYCNT .. H_ARMYCNT+MAXP-1
      const int lane = lane_id();
      const uint32_t tot = multi_sum(acc);
      hv = (lane >= H_ARMYCNT && lane < H_ARMYCNT + MAXP) ? tot : hv;
    }
    alive = 0u;
#pragma unroll
    for (int p = 0; p < MAXP; ++p) {
      // GeneralIdx: the reference keeps the last general in list order (:46,:101,:122); with two
      // or more generals that order depends on Go map iteration.  Here: the highest tile index.
      const uint32_t g = lst[p] & gen;
      const unsigned long long dw = __builtin_amdgcn_ballot_w64(g != 0u);
      int gi = -1;
      if (dw) {
        const int i = 63 - __builtin_clzll(dw);
        gi = 32 * i + (31 - __builtin_clz(rdlane(g, i)));
      }
      hdr_set(H_GIDX + p, (uint32_t)gi);
      alive |= (gi >= 0) ? (1u << p) : 0u;  // :52-54 / :133-135
    }
  }

  // ---- ProductionManager.ProcessTurnProduction (production_manager.go:26-101) -------------
  __device__ __forceinline__ void production(int pg, int pc, int pn, int interval) {
    const bool grow = (turn % interval) == 0;  // :27
    uint32_t listed_alive = 0u;                // :39-45: lists of alive players, owner NOT re-checked (H7)
#pragma unroll
    for (int p = 0; p < MAXP; ++p) listed_alive |= ((alive >> p) & 1u) ? lst[p] : 0u;
    const uint32_t normal = ~(gen | city | mtn) & valid;
    const uint32_t mg = (pg > 0) ? (listed_alive & gen) : 0u;
    const uint32_t mc = (pc > 0) ? (listed_alive & city) : 0u;
    const uint32_t mn = (grow && pn > 0) ? (listed_alive & normal) : 0u;
    chg |= mg | mc | mn;  // :59-61 (prod > 0 only)
    const int an = (grow && pn > 0) ? pn : pc;
    if (pg == pc && pc == an) {  // one rate for every producing tile: one gather per slot
      const uint32_t m = mg | mc | mn;
#pragma unroll
      for (int s = 0; s < NSLOT; ++s) army[s] = (int32_t)(__umul24(gather(m, s), (uint32_t)pg) + (uint32_t)army[s]);  // rates < 2^24 (gvec_create)
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
    // the winner is re-derived from Alive on read-back, like Engine.GetWinner (engine.go:248-263)
  }

  // ---- the action phase: ActionProcessor.ProcessActions (processor/action_processor.go:36-99) ----
  // Everything about a move that does not depend on the board as modified by lower player ids is
  // computed for all players at once on lanes (lane p = player p): coordinate unpack, bounds,
  // same-tile and adjacency checks of MoveAction.Validate (core/action.go:58-76), tile indices.
  struct ActVec {
    uint32_t meta;  // bits 0-3 static Validate code, bit 4 present (non-nil), bit 5 half
    int ft, tt;     // y*W + x of source / target
  };
  __device__ __forceinline__ ActVec prevalidate(uint32_t alo, uint32_t ahi) const {
    ActVec v;
    const int fx = (int)(int8_t)(alo & 0xFFu), fy = (int)(int8_t)((alo >> 8) & 0xFFu);
    const int tx = (int)(int8_t)((alo >> 16) & 0xFFu), ty = (int)(int8_t)(alo >> 24);
    const bool inb = fx >= 0 && fx < W && fy >= 0 && fy < H && tx >= 0 && tx < W && ty >= 0 && ty < H;
    const int dx = fx - tx, dy = fy - ty;
    const int md = (dx < 0 ? -dx : dx) + (dy < 0 ? -dy : dy);
    uint32_t code = 0u;
    code = (md != 1) ? GVEC_ERR_NOT_ADJACENT : code;      // action.go:72-76 (orthogonal, one step)
    code = (md == 0) ? GVEC_ERR_MOVE_TO_SELF : code;      // :67-69
    code = (!inb) ? GVEC_ERR_INVALID_COORDINATES : code;  // :58-64
    v.meta = code | ((ahi & GVEC_ACT_VALID) ? 16u : 0u) | ((ahi & GVEC_ACT_HALF) ? 32u : 0u);
    v.ft = __mul24(fy, W) + fx;  // |coordinates| < 128: full-rate 24-bit multiply
    v.tt = __mul24(ty, W) + tx;
    return v;
  }

  // The state-dependent rest of Validate + core.ApplyMoveAction (core/movement.go:23-89) for
  // player PID, wave-uniform.  `orders` packs core.ProcessCaptures' output (movement.go:100-118):
  // byte k = victim | new_owner << 4.
  template <int PID>
  __device__ __forceinline__ void apply_action(const ActVec& av, uint32_t& first_err, uint64_t& orders, int& n_orders,
                                               uint32_t& elim_seen) {
    const uint32_t m = rdlane(av.meta, PID);
    if (!(m & 16u)) return;              // nil action
    if (!((alive >> PID) & 1u)) return;  // action_processor.go:56-60 (Alive as last written, H2)
    uint32_t code = m & 15u;
    int fa = 0, ta = 0, ft = 0, tt = 0;
    if (!code) {
      ft = (int)rdlane((uint32_t)av.ft, PID);
      tt = (int)rdlane((uint32_t)av.tt, PID);
      fa = army_get(ft);
      ta = army_get(tt);
      if (!bit_at(own[PID], ft)) code = GVEC_ERR_NOT_OWNED;          // action.go:82-84
      else if (fa <= 1) code = GVEC_ERR_INSUFFICIENT_ARMY;           // :87-89
      else if (bit_at(mtn, tt)) code = GVEC_ERR_TARGET_IS_MOUNTAIN;  // :96-98
    }
    if (code) {  // action_processor.go:66-77: remember the FIRST error, keep going
      first_err = first_err ? first_err : code;
      return;
    }
    int n = (m & 32u) ? (fa / 2) : (fa - 1);  // movement.go:40-49
    n = (n == 0) ? 1 : n;
    const uint32_t fbit = tile_bit(ft), tbit = tile_bit(tt);
    army_set(ft, fa - n);  // :54
    chg |= fbit | tbit;    // :57-60
    if (bit_at(own[PID], tt)) {  // :62-66 own tile: consolidate
      army_set(tt, ta + n);
    } else if (n > ta) {  // :69-82 capture (ties favour the defender)
      int prev = -1;
#pragma unroll
      for (int q = 0; q < MAXP; ++q) {
        if (bit_at(own[q], tt)) prev = q;
        own[q] &= ~tbit;
      }
      own[PID] |= tbit;
      army_set(tt, n - ta);
      vch |= tbit;  // action_processor.go:84-86
      // movement.go:105-108
      if (bit_at(gen, tt) && prev >= 0 && !((elim_seen >> prev) & 1u)) {
        orders |= (uint64_t)((uint32_t)prev | ((uint32_t)PID << 4)) << (8 * n_orders);
        n_orders++;
        elim_seen |= 1u << prev;
      }
    } else {
      army_set(tt, ta - n);  // :85
    }
  }

  template <int PID>
  __device__ __forceinline__ void act_chain(const ActVec& av, uint32_t& first_err, uint64_t& orders, int& n_orders,
                                            uint32_t& elim_seen) {
    if constexpr (PID < MAXP) {
      if (PID < P) apply_action<PID>(av, first_err, orders, n_orders, elim_seen);
      act_chain<PID + 1>(av, first_err, orders, n_orders, elim_seen);
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
      }
      hdr_set(H_GIDX + v, 0xFFFFFFFFu);  // :141 GeneralIdx = -1
      chg |= tiles;                      // :133-134
      vch |= tiles;
      alive &= ~(1u << v);  // :140
    }
  }

  // ---- TurnProcessor.ProcessTurn (turn_processor.go:29-77) ----------------------------------
  // acts_lo/hi: lane p holds player p's gvec_action words.  Returns the per-env error code.
  // Precondition: the caller has checked Engine.gameOver (validateGameState :95-113).
  __device__ __forceinline__ uint32_t turn_step(uint32_t acts_lo, uint32_t acts_hi, const StepArgs& A, bool& aborted) {
    aborted = false;
    turn++;  // initializeTurn :124-135
    update_fog();
    chg = 0u;
    vch = 0u;
    uint32_t first_err = 0u, elim_seen = 0u;
    uint64_t orders = 0ull;
    int n_orders = 0;
    // Engine.processActions (engine.go:80-115): PlayerID order == slot order (sort.Slice :39-41)
    const ActVec av = prevalidate(acts_lo, acts_hi);
    const unsigned long long present = __builtin_amdgcn_ballot_w64((av.meta & 16u) != 0u && lane_id() < P);
    if (present) {  // a turn where nobody moves touches no army
      army_to_lds();
      act_chain<0>(av, first_err, orders, n_orders, elim_seen);
      army_from_lds();
    }
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
  // A player's packed mask is FOUR DIRECTION BIT-PLANES of fd dwords each: bit t of plane d = action
  // (y*W+x)*4 + d of the reference, t = y*W + x.  Plane d is then just
  //     (listed & owned & army > 1)  &  ("the neighbour in direction d is on the board, not a mountain")
  // in the flat domain - two ANDs - and the row [d][i] is laid on lanes j = d*fd + i with one
  // ds_bpermute per plane (lane j reads flat lane j mod fd), so that lane j stores dword j: one
  // coalesced store per player.  out[p][k]: lane l holds dword l + 64k of player p's row.
  // SERIALIZER = false: Engine.GetLegalActionMask, d = 0 up, 1 right, 2 down, 3 left (H10).
  // SERIALIZER = true : Serializer.GenerateActionMask (internal/experience/serializer.go:112-176):
  //   board owner (not the list), army >= 2, no Alive check, d = 0 up, 1 DOWN, 2 LEFT, 3 right (H10).
  // The player-independent half of the mask, laid out like a mask row: lane j (+64k) holds, for
  // direction plane d = j / fd, the dword j % fd of "the d-neighbour is on the board and not a
  // mountain".  Depends on the type planes and the geometry only: constant for the life of a board.
  template <bool SERIALIZER = false>
  __device__ __forceinline__ void legal_targets(uint32_t (&okp)[MPASS], int fd) const {
    const uint32_t notm = ~mtn & valid;  // in-board, not a mountain (Validate :58-64,:96-98)
    const uint32_t ok_up = upW(notm), ok_dn = dnW(notm);                  // target y-1 / y+1
    const uint32_t ok_rt = dn1(notm) & ncolL, ok_lf = up1(notm) & ncol0;  // target x+1 / x-1
    const uint32_t ok1 = SERIALIZER ? ok_dn : ok_rt, ok2 = SERIALIZER ? ok_lf : ok_dn, ok3 = SERIALIZER ? ok_rt : ok_lf;
#pragma unroll
    for (int k = 0; k < MPASS; ++k) {
      const int j = lane_id() + 64 * k;
      const int d = (j >= fd ? 1 : 0) + (j >= 2 * fd ? 1 : 0) + (j >= 3 * fd ? 1 : 0);
      const int addr = (j - d * fd) << 2;  // flat lane j mod fd
      // cross-lane reads: all unconditional
      const uint32_t g0 = bperm(addr, ok_up), g1 = bperm(addr, ok1), g2 = bperm(addr, ok2), g3 = bperm(addr, ok3);
      const uint32_t g = (d == 0) ? g0 : (d == 1) ? g1 : (d == 2) ? g2 : g3;
      okp[k] = (j < 4 * fd) ? g : 0u;
    }
  }
  template <bool SERIALIZER = false>
  __device__ __forceinline__ void legal_masks(uint32_t (&out)[MAXP][MPASS], int fd, const uint32_t (&okp)[MPASS]) const {
    uint32_t gt1 = 0u;  // army > 1 as a flat plane
#pragma unroll
    for (int s = 0; s < NSLOT; ++s) scatter(gt1, __builtin_amdgcn_ballot_w64(army[s] > 1), s);
    uint32_t src[MAXP];
#pragma unroll
    for (int p = 0; p < MAXP; ++p) {
      // legal_moves.go :26-28 alive, :37 listed, :41 owner == pid && army > 1
      const uint32_t m = SERIALIZER ? (own[p] & gt1) : (lst[p] & own[p] & gt1);
      src[p] = (SERIALIZER || ((alive >> p) & 1u)) ? m : 0u;
    }
#pragma unroll
    for (int k = 0; k < MPASS; ++k) {
      const int j = lane_id() + 64 * k;
      const int d = (j >= fd ? 1 : 0) + (j >= 2 * fd ? 1 : 0) + (j >= 3 * fd ? 1 : 0);
      const int addr = (j - d * fd) << 2;
#pragma unroll
      for (int p = 0; p < MAXP; ++p) out[p][k] = bperm(addr, src[p]) & okp[k];  // cross-lane: unconditional
    }
  }
  template <bool SERIALIZER = false>
  __device__ __forceinline__ void legal_masks(uint32_t (&out)[MAXP][MPASS], int fd) const {
    uint32_t okp[MPASS];
    legal_targets<SERIALIZER>(okp, fd);
    legal_masks<SERIALIZER>(out, fd, okp);
  }

  // ---- internal/experience/rewards.go helpers ---------------------------------------------------
  // sum of Tile.Army over a flat plane (countPlayerArmies, rewards.go:98-107)
  __device__ __forceinline__ int32_t army_sum(uint32_t plane) const {
    int32_t acc = 0;
#pragma unroll
    for (int s = 0; s < NSLOT; ++s) acc += army[s] & gather_mask(plane, s);
    return (int32_t)wave_sum((uint32_t)acc);
  }
  __device__ __forceinline__ int32_t count(uint32_t plane) const { return (int32_t)wave_sum((uint32_t)__builtin_popcount(plane)); }
};

}  // namespace gvec
