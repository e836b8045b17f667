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
//       gt1     Tile.Army > 1 of the resident state (the army-dependent half of every legal-move
//               predicate: stored so the next launch need not rebuild it from the armies)
//       valid / ncol0 / ncolL / ok[4]   geometry masks and "the d-neighbour is on the board and not a
//               mountain": functions of the board size and the type planes only, computed once when a
//               board is imported and carried as constant planes (loading 364 bytes costs less than
//               the ~55 vector instructions that rebuild them: the step kernel is VALU-issue bound)
//
// Planes and armies are loaded straight into registers (plane p: lane i reads dword i; armies:
// 256 B per wave instruction); the planes block of one env is a contiguous, 16-byte aligned run.  Integer / bit work only: no MFMA
// anywhere (HBM-roofline kernel).
//
// This file holds the PLAIN layout (`Board`: one register per plane, used by the conversion and experience
// kernels) and everything both layouts share; the turn logic lives in gvec_packed.hpp (`PBoard`).
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

// Streaming (non-temporal) cache policy on the turn kernels' once-per-launch accesses, by class (bits):
// 1 loads, 2 mask stores, 4 plane stores, 8 army stores.  One-process A/B at 262,144 boards 20x20 4P: nt LOADS
// cost 12 %, nt stores gain 3-4 % (the stored lines need not displace what the next loads want).
#ifndef GVEC_NT
#define GVEC_NT 14
#endif
#define GVEC_NT_MASK 2
#define GVEC_NT_PLANE 4
#define GVEC_NT_ARMY 8
// 1: the streaming stores carry `sc1 nt` - written through, the line dropped from the XCD's L2 - instead of `nt` alone.
// scripts/microbench/copy_pattern2.hip, the step kernel's bytes as 16-byte-per-lane stores: nt 255 us, nt sc1 196 us,
// sc1 alone 290 us, no bits 280 us per 262,144 boards.  The compiler has no spelling for the pair short of `volatile`
// (which also serialises the stores), hence the inline assembly; nothing ever waits for these stores, and the one
// wait state a wide store's data registers need before they are rewritten is in the string.
#ifndef GVEC_SC1
#define GVEC_SC1 1
#endif

namespace gvec {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <typename T>
__device__ __forceinline__ T ld_stream(const T* p) {
  if constexpr ((GVEC_NT & 1) != 0) return __builtin_nontemporal_load(p);
  return *p;
}
__device__ __forceinline__ void st_through(uint32_t* p, uint32_t v) { asm volatile("global_store_dword %0, %1, off sc1 nt" : : "v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void st_through(uint16_t* p, uint16_t v) {
  asm volatile("global_store_short %0, %1, off sc1 nt" : : "v"(p), "v"((uint32_t)v) : "memory");
}
__device__ __forceinline__ void st_through(float* p, float v) { asm volatile("global_store_dword %0, %1, off sc1 nt" : : "v"(p), "v"(v) : "memory"); }
// gfx940-family hazard: a VMEM store of more than 64 bits needs 2 wait states before a VALU write of its data
// registers, and the compiler's hazard recognizer cannot see a store inside an asm string: `s_nop 1` (= 2 wait
// states) is the only protection.  tests/test_kernel_asm.py scans the generated code for every such store.
__device__ __forceinline__ void st_through(u32x4* p, u32x4 v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc1 nt\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
}
template <int CLASS, typename T>
__device__ __forceinline__ void st_stream(T* p, T v) {
  if constexpr ((GVEC_NT & CLASS) != 0) {
    if constexpr (GVEC_SC1 != 0) st_through(p, v);
    else __builtin_nontemporal_store(v, p);
  } else {
    *p = v;
  }
}

// ---- resident record layout (per env) -------------------------------------------------
// hdr  : HDR_DW u32      planes : M*FD u32 (+pad to 4), plane-major [m][i]
// army : NARROW block, NSLOT*64 u16 (the default), or - header flag HF_WIDE - the env's block of the
//        int32 escape array, NSLOT*64 i32 (see "army storage" below)
constexpr int HDR_DW = 24;
enum : int {
  H_TURN = 0,     // GameState.Turn
  H_DIMS = 1,     // W | H<<8 | P<<16 | flags<<24   (flags: HF_* below)
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
constexpr uint32_t HF_SETUP = 8u;  // transient: imported with init, performInitialSetup still to run (setup_kernel clears it)
// Bookkeeping of the turn engine (never part of the exported state; all three absent = the general paths):
constexpr uint32_t HF_SYNC = 16u;        // the player stats are exactly what a stats pass over the current lists and armies yields
                                         // (set by every pass, cleared by an aborted turn and by every state import): the next
                                         // pass may add the turn's deltas to ArmyCount instead of summing the board
constexpr uint32_t HF_VSMALL = 32u;      // VisibilityChangedTiles holds at most one tile per player (no turnover since it was cleared)
constexpr uint32_t HF_LDIFF = 128u;      // some player's OwnedTiles differ from what it owns (H6: after an aborted turn, until a pass heals
                                         // it): only then are the LST planes of the block authoritative - and read or written at all.
                                         // Clear: lists == ownership, the stored LST planes are stale.  Kept true by settle_lists().
constexpr uint32_t HF_FEWSPECIAL = 64u;  // 2*P + generals + cities <= N/5 (a function of the board): without growth or turnover a
                                         // turn's ChangedTiles cannot reach the full-pass threshold

// plane order inside the planes block: what every turn rewrites (OWN .. GT1, contiguous), what changes only when a
// board is imported or re-dealt (GEN .. OK[3]), and last the list planes, which most turns neither read nor write
// (HF_LDIFF)
template <int MAXP>
struct Planes {
  static constexpr int OWN = 0, VIS = MAXP, CHG = 2 * MAXP, VCH = 2 * MAXP + 1, GT1 = 2 * MAXP + 2,
                       GEN = 2 * MAXP + 3, CITY = 2 * MAXP + 4, MTN = 2 * MAXP + 5, VALID = 2 * MAXP + 6, NCOL0 = 2 * MAXP + 7,
                       NCOLL = 2 * MAXP + 8, OK = 2 * MAXP + 9 /* [4]: up, right, down, left */, LST = 2 * MAXP + 13,
                       COUNT = 3 * MAXP + 13, MUTABLE = 2 * MAXP + 3, SHARED = 13 /* CHG .. OK[3] */;
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
  const uint32_t* zeros;  // >= row_dw zero dwords: what the lanes outside a board's bit string load
  const uint32_t* pool_hdr;
  const uint32_t* pool_rows;
  const uint32_t* pool_army16;
  const int32_t* pool_army32;
  int32_t num_envs, fd, row_dw, mask_dw, pool_size;  // fd = dwords per plane = ceil(max tiles / 32)
  int32_t pstride;  // players per env in actions / legal buffers (gvec_config.max_players)
  int32_t prod_general, prod_city, prod_normal, interval;
  uint32_t interval_magic;  // ceil(2^32 / interval)
  int32_t turns, invalid_permille;
  uint32_t agent_noop, agent_half;  // gvec_set_agent_mix thresholds (of 65536)
  uint32_t flags, seed_lo, seed_hi, pool_seed_lo, pool_seed_hi;
  uint32_t seed_base, pool_seed_base;  // env_key_base of the two seeds: filled in by the launchers (with_seed_bases)
  int32_t env_base;                    // host side only: this handle's first env within its sharded batch (folded into the bases)
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
    const uint32_t w = ld_stream(n + 64 * k + lane);
    army[2 * k] = (int32_t)(w & 0xFFFFu);
    army[2 * k + 1] = (int32_t)(w >> 16);
  }
  if constexpr ((NSLOT & 1) != 0) army[NSLOT - 1] = (int32_t)ld_stream(reinterpret_cast<const uint16_t*>(n + 64 * (NSLOT / 2)) + lane);
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
  for (int k = 0; k < NSLOT / 2; ++k) st_stream<GVEC_NT_ARMY>(n + 64 * k + lane, (uint32_t)army[2 * k] | ((uint32_t)army[2 * k + 1] << 16));
  if constexpr ((NSLOT & 1) != 0) st_stream<GVEC_NT_ARMY>(reinterpret_cast<uint16_t*>(n + 64 * (NSLOT / 2)) + lane, (uint16_t)army[NSLOT - 1]);
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
__device__ __forceinline__ uint64_t uni64(uint64_t v) {
  return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
}
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
__host__ __device__ __forceinline__ uint32_t fmix32(uint32_t h) {
  h ^= h >> 16;
  h *= 0x85EBCA6Bu;
  h ^= h >> 13;
  h *= 0xC2B2AE35u;
  h ^= h >> 16;
  return h;
}
// env_key(lo, hi, env) = env_key_of(env_key_base(lo, hi), env): the base is a function of the launch's seed alone and
// is hashed once on the host (StepArgs::seed_base), not once per board on the scalar unit
__host__ __device__ __forceinline__ uint32_t env_key_base(uint32_t lo, uint32_t hi) { return fmix32(lo ^ 0x9E3779B9u) + hi * 0x85EBCA77u + 0x27D4EB2Fu; }
__host__ __device__ __forceinline__ uint32_t env_key_of(uint32_t base, uint32_t env) { return fmix32(base + env * 0xC2B2AE3Du); }
__host__ __device__ __forceinline__ uint32_t env_key(uint32_t lo, uint32_t hi, uint32_t env) { return env_key_of(env_key_base(lo, hi), env); }

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
// Board: the PLAIN register layout - one register per (plane kind, player); lane i holds dword i of the bit
// string.  Used by the kernels that convert between the resident record and per-tile planes (import /
// export / records) and by the experience kernels; it owns the routines that BUILD the constant planes.
// =========================================================================================
template <int MAXP, int NSLOT>
struct Board {
  using PL = Planes<MAXP>;

  // flat domain
  uint32_t own[MAXP], lst[MAXP], vis[MAXP], chg, vch, gt1, gen, city, mtn;
  uint32_t valid;  // bits t < N
  uint32_t ncol0;  // bits with x != 0      (guards a shift towards higher t)
  uint32_t ncolL;  // bits with x != W - 1  (guards a shift towards lower t)
  uint32_t ok[4];  // the neighbour in direction d (0 up, 1 right, 2 down, 3 left) is on the board and not a mountain
  // tile domain
  int32_t army[NSLOT];
  // header: lane k holds header dword k
  uint32_t hv;
  // wave-uniform
  int W, H, P, N, turn, recipW;
  uint32_t alive, hflags;

  // ---- geometry masks of this board size (computed on import, then carried as planes) ---------
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
    ncol0 = (lane_id() < 32) ? ncol0 : 0xFFFFFFFFu;                // lanes beyond the bit string hold no tiles: any value
    ncolL = (lane_id() < 32) ? ncolL : 0xFFFFFFFFu;                // would do, a fixed one keeps the stored planes canonical
  }
  // the player-independent half of MoveAction.Validate (core/action.go:58-64, :96-98) per direction,
  // rules/legal_moves.go:13-18 order: needs geometry() and the type planes
  __device__ __forceinline__ void targets() {
    const uint32_t notm = ~mtn & valid;
    ok[0] = upW(notm);          // target (x, y-1)
    ok[1] = dn1(notm) & ncolL;  // target (x+1, y)
    ok[2] = dnW(notm);          // target (x, y+1)
    ok[3] = up1(notm) & ncol0;  // target (x-1, y)
  }
  // the header flags that are functions of the board, and "nothing is known" for the turn engine's bookkeeping
  // flags: wherever a board is imported or modified from outside
  __device__ __forceinline__ void static_flags() {
    const int special = (int)wave_sum((uint32_t)__builtin_popcount(gen | city));
    hflags = (hflags & ~(HF_SYNC | HF_VSMALL | HF_FEWSPECIAL)) | ((2 * P + special <= N / 5) ? HF_FEWSPECIAL : 0u);
  }
  // Tile.Army > 1 as a flat plane
  __device__ __forceinline__ void refresh_gt1() {
    gt1 = 0u;
#pragma unroll
    for (int s = 0; s < NSLOT; ++s) scatter(gt1, __builtin_amdgcn_ballot_w64(army[s] > 1), s);
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

  __device__ __forceinline__ void load_planes(const uint32_t* rows_env, int fd) {
    const int lane = lane_id();
    const bool on = lane < fd;
    const uint32_t* g = rows_env + (on ? lane : 0);
    auto ld = [&](int plane) {
      const uint32_t v = g[plane * fd];
      return on ? v : 0u;
    };
    const bool listed = (hflags & HF_LDIFF) != 0u;  // needs the header: call after load_hdr
#pragma unroll
    for (int p = 0; p < MAXP; ++p) {
      own[p] = ld(PL::OWN + p);
      vis[p] = ld(PL::VIS + p);
      lst[p] = own[p];
      if (listed) lst[p] = ld(PL::LST + p);
    }
    chg = ld(PL::CHG);
    vch = ld(PL::VCH);
    gt1 = ld(PL::GT1);
    gen = ld(PL::GEN);
    city = ld(PL::CITY);
    mtn = ld(PL::MTN);
    valid = ld(PL::VALID);
    ncol0 = ld(PL::NCOL0);
    ncolL = ld(PL::NCOLL);
#pragma unroll
    for (int d = 0; d < 4; ++d) ok[d] = ld(PL::OK + d);
  }

  // Sets HF_LDIFF to what the lists are: call BEFORE store_hdr (store_planes writes the list planes accordingly).
  __device__ __forceinline__ void settle_lists() {
    uint32_t d = 0u;
#pragma unroll
    for (int p = 0; p < MAXP; ++p) d |= lst[p] ^ own[p];
    hflags = wave_any(d != 0u) ? (hflags | HF_LDIFF) : (hflags & ~HF_LDIFF);
  }
  // The planes from GEN on change only when a board is imported or re-dealt (with_types).  all_lists: write the list
  // planes whatever the flag says (a record slab is complete on its own).
  __device__ __forceinline__ void store_planes(uint32_t* rows_env, int fd, int row_dw, bool with_types, bool all_lists = false) const {
    const int lane = lane_id();
    if (lane < fd) {
      uint32_t* g = rows_env + lane;
#pragma unroll
      for (int p = 0; p < MAXP; ++p) {
        g[(PL::OWN + p) * fd] = own[p];
        g[(PL::VIS + p) * fd] = vis[p];
        if (all_lists || (hflags & HF_LDIFF)) g[(PL::LST + p) * fd] = lst[p];
      }
      g[PL::CHG * fd] = chg;
      g[PL::VCH * fd] = vch;
      g[PL::GT1 * fd] = gt1;
      if (with_types) {
        g[PL::GEN * fd] = gen;
        g[PL::CITY * fd] = city;
        g[PL::MTN * fd] = mtn;
        g[PL::VALID * fd] = valid;
        g[PL::NCOL0 * fd] = ncol0;
        g[PL::NCOLL * fd] = ncolL;
#pragma unroll
        for (int d = 0; d < 4; ++d) g[(PL::OK + d) * fd] = ok[d];
      }
    }
    if (with_types && lane < row_dw - PL::COUNT * fd) rows_env[PL::COUNT * fd + lane] = 0u;  // the block's padding
  }

  // whole 64-tile slots travel both ways (the padding beyond N holds zeros).
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

  // ---- flat <-> tile domain ------------------------------------------------------------------
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
    plane = (uint32_t)gvec_llvm_writelane((int)(uint32_t)ballot, 2 * s, (int)plane);
    plane = (uint32_t)gvec_llvm_writelane((int)(uint32_t)(ballot >> 32), 2 * s + 1, (int)plane);
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
