// gvec_packed.hpp — the turn engine's register layout: per-player planes packed into register ROWS.
//
// Board (gvec_device.hpp) keeps one register per (plane kind, player), of which a 20x20 board uses 13 of
// 64 lanes.  PBoard lays the players of one kind side by side in one register:
//
//     row r of register k (lanes r*ROWL .. r*ROWL + ROWL-1)  =  player k*PPR + r
//     lane column c = lane % ROWL                            =  dword c of the bit string (tiles 32c..32c+31)
//
// with ROWL = 16 (PPR = 4 players per register) for boards up to 448 tiles and ROWL = 32 (PPR = 2) above.
// The planes every player shares (chg, vch, gt1, the type planes, the geometry masks and the four
// "target is free" planes) are REPLICATED in every row, so a packed plane combines with them lane by lane.
// Every per-player flat operation of the turn (fog dilations, list refresh, the legal-move planes, the
// random agent's prefix scan) runs once per register instead of once per player.  The wave-wide bit-string
// shifts stay correct because the last lanes of every row hold no tiles (fd <= 14 of 16, fd <= 32 of 32 -
// the one full case is masked explicitly).
//
// All turn logic lives here (TurnProcessor.ProcessTurn and everything below it, the legal-action mask,
// performInitialSetup); every routine cites the Go function it reproduces, paths relative to
// /root/reference/internal/game/.  HBM layout, tile domain (armies) and header are Board's.
#pragma once
#include "gvec_device.hpp"

// Profiling builds only (scripts/profile_phases.sh): -DGVEC_PROFILE_SKIP=<bits> compiles one phase of the turn out
// so that per-phase instruction counts can be read off the PMC counters.  Never defined in the shipped library.
#ifndef GVEC_PROFILE_SKIP
#define GVEC_PROFILE_SKIP 0
#endif
// -DGVEC_PROFILE_DUP=<bits> runs an idempotent phase TWICE instead (same bits; the game is unchanged, so the counter
// difference to the plain build is that phase's dynamic cost - skipping a phase changes what the boards do next)
#ifndef GVEC_PROFILE_DUP
#define GVEC_PROFILE_DUP 0
#endif
// 1: the action phase runs on lanes when no two moves of a turn share a tile (act_vector); 0: always sequential (A/B)
#ifndef GVEC_ACT_VECTOR
#define GVEC_ACT_VECTOR 1
#endif

namespace gvec {

// ---- row-wise cross-lane helpers (a "row" = ROWL consecutive lanes) ---------------------------------
template <int ROWL>
__device__ __forceinline__ uint32_t row_scan_add(uint32_t v) {  // inclusive prefix sum inside every row
  v += dpp0<0x111>(v);  // row_shr:1 (DPP rows are 16 lanes: zeros are shifted in at their start)
  v += dpp0<0x112>(v);
  v += dpp0<0x114>(v);
  v += dpp0<0x118>(v);
  if constexpr (ROWL == 32) v += dpp0<0x142, 0xa>(v);  // row_bcast15 into DPP rows 1 and 3
  return v;
}
template <int ROWL>
__device__ __forceinline__ uint32_t row_scan_or(uint32_t v) {  // inclusive prefix OR inside every row
  v |= dpp0<0x111>(v);
  v |= dpp0<0x112>(v);
  v |= dpp0<0x114>(v);
  v |= dpp0<0x118>(v);
  if constexpr (ROWL == 32) v |= dpp0<0x142, 0xa>(v);
  return v;
}
// every lane receives the value of its row's LAST lane (ds_swizzle bit mode: lane = (lane & and) | or)
template <int ROWL>
__device__ __forceinline__ uint32_t row_last(uint32_t v) {
  if constexpr (ROWL == 16) return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, 0x10 | (0x0F << 5));
  return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, 0x1F << 5);
}

// position of the r-th (0-based) set bit of w, per lane (r < popcount(w), else unspecified)
__device__ __forceinline__ uint32_t kth_set_bit(uint32_t w, uint32_t r) {
  uint32_t pos = 0u;
#pragma unroll
  for (int width = 16; width >= 2; width >>= 1) {
    const uint32_t c = (uint32_t)__builtin_popcount(__builtin_amdgcn_ubfe(w, pos, (uint32_t)width));
    const uint32_t d = r - c;          // borrows iff r < c
    const bool lt = r < c;
    r = lt ? r : d;
    pos = lt ? pos : (pos | (uint32_t)width);   // pos is a multiple of 2 * width: OR == ADD
  }
  // last level: the bit at pos is the one iff it is set and r == 0
  return pos | ((r >= ((w >> pos) & 1u)) ? 1u : 0u);
}

// ---- the random agent's mixer (DESIGN.md "Synthetic inputs"; mirrored by the oracle's ora_amix) --------
// Two rounds of xorshift + 24-bit multiply: v_mul_u32_u24 is full rate, fmix32's v_mul_lo_u32 costs four
// issue slots each (the step kernel is VALU-issue bound: three fmix32 per env-step were 5 % of it).
__device__ __forceinline__ uint32_t amix(uint32_t x) {
  x ^= x >> 15;
  x = __umul24(x, 0xE8A54Du);
  x ^= x >> 13;
  x = __umul24(x, 0xAA34A7u);
  x ^= x >> 15;
  return x;
}

template <int MAXP, int NSLOT>
struct PBoard {
  using PL = Planes<MAXP>;
  static constexpr int PPR = (NSLOT <= 7) ? 4 : 2;    // players per plane register
  static constexpr int ROWL = 64 / PPR;               // lanes per row
  static constexpr int NR = (MAXP + PPR - 1) / PPR;   // registers per plane kind
  static constexpr bool FULL_ROWS = (2 * NSLOT >= ROWL);  // a plane can fill its row: shifts must not cross rows
  // LDS scratch of the vector action phase: three ROWL-dword tile bitmaps (candidate tiles, changed, captured), one
  // 64-dword image per packed ownership register (the bits captures set) and 8 dwords of army lost per defending player
  // ... and 64 dwords where the lanes that have nothing to write put it (an address select instead of an exec-mask region:
  // the scalar unit, not the vector unit, is the busy one), the whole rounded to 64 so that clearing it needs no mask
  static constexpr int ACT_DUMMY = 3 * ROWL + 64 * NR + 8;
  static constexpr int ACT_SCRATCH_DW = (ACT_DUMMY + 64 + 63) / 64 * 64;

  uint32_t own[NR], lst[NR], vis[NR];         // packed: row r of register k = player k*PPR + r
  uint32_t chg, vch, gt1, gen, city, mtn;     // replicated in every row
  uint32_t valid, ncol0, ncolL, ok[4];        // replicated constant planes (see Board)
  uint32_t rowbit[NR];                        // 1 << (the player this lane holds in register k)
  static constexpr int NSHARED = PL::SHARED;                    // the 13 shared planes, CHG .. OK[3], contiguous in the block
  static constexpr int NSR = (NSHARED + PPR - 1) / PPR;         // ... loaded PPR at a time, like the packed ones
  uint32_t shp[NSR];                          // in flight between load_planes and spread_shared: row r of shp[k] = shared plane k*PPR + r
  int32_t army[NSLOT];                        // tile domain, as in Board
  uint32_t hv;
  int32_t* larmy;                             // LDS shadow of the armies during the action phase: tile t at larmy[t]
  uint32_t* lscr = nullptr;                   // ACT_SCRATCH_DW dwords of LDS for the vector action phase (null: sequential only)
  int W, H, P, N, turn, recipW;
  uint32_t alive, hflags;
  // every army stays below 2^23 until the board is stored (true for one turn from a NARROW load): the
  // masked sums may then use full-rate 24-bit multiply-adds.  Set by the kernel, never by load_*.
  bool small = false;
  // what one turn hands from its action and production phases to the end-of-turn stats pass (update_stats)
  int32_t move_delta = 0;      // lane p: change of player p's listed army through this turn's moves (act_vector)
  uint32_t prod_mask = 0u;     // the tiles production raised, replicated in every row ...
  int32_t prod_rate = 0;       // ... by this much each (prod_uniform), else the delta path is off
  bool prod_uniform = false, grow = false;

  static __device__ __forceinline__ int col() { return lane_id() & (ROWL - 1); }
  static __device__ __forceinline__ int row() { return lane_id() / ROWL; }
  // the player whose bits this lane holds in register k
  static __device__ __forceinline__ int lane_player(int k) { return k * PPR + row(); }
  // bit (k*PPR + row) of a per-player bit set: "the property holds for this lane's player"
  __device__ __forceinline__ bool lane_flag(uint32_t bits, int k) const { return (bits & rowbit[k]) != 0u; }
  static __device__ __forceinline__ bool in_row_of(int p) { return row() == (p % PPR); }

  // ---- bit-string shifts inside a row -------------------------------------------------------------
  static __device__ __forceinline__ uint32_t prev_lane(uint32_t m) {
    const uint32_t v = from_prev(m);
    if constexpr (FULL_ROWS) return (col() == 0) ? 0u : v;
    return v;  // the previous row's last lanes hold no tiles
  }
  static __device__ __forceinline__ uint32_t next_lane(uint32_t m) {
    const uint32_t v = from_next(m);
    if constexpr (FULL_ROWS) return (col() == ROWL - 1) ? 0u : v;
    return v;  // lands in a lane beyond the board (masked by `valid`) or reads a zero lane
  }
  __device__ __forceinline__ uint32_t up1(uint32_t m) const { return __builtin_amdgcn_alignbit(m, prev_lane(m), 31); }
  __device__ __forceinline__ uint32_t dn1(uint32_t m) const { return __builtin_amdgcn_alignbit(next_lane(m), m, 1); }
  __device__ __forceinline__ uint32_t upW(uint32_t m) const {
    return __builtin_amdgcn_alignbit(m, prev_lane(m), (uint32_t)(32 - W) & 31u);
  }
  __device__ __forceinline__ uint32_t dnW(uint32_t m) const { return (uint32_t)((((uint64_t)next_lane(m) << 32) | (uint64_t)m) >> W); }
  __device__ __forceinline__ uint32_t dil_h(uint32_t m) const { return m | (up1(m) & ncol0) | (dn1(m) & ncolL); }
  __device__ __forceinline__ uint32_t dil_v(uint32_t m) const { return (m | upW(m) | dnW(m)) & valid; }
  // 3x3 / 5x5 neighbourhoods (visibility_optimized.go:9-13, :104-105)
  __device__ __forceinline__ uint32_t dil3(uint32_t m) const { return dil_v(dil_h(m)); }

  // OR of the rows of a packed plane, replicated into every row
  static __device__ __forceinline__ uint32_t or_rows(uint32_t x) {
    if constexpr (PPR == 4) x |= (uint32_t)__builtin_amdgcn_ds_swizzle((int)x, 0x401F);  // lane ^ 16
    x |= bperm((lane_id() ^ 32) << 2, x);
    return x;
  }

  // ---- header (Board's) ---------------------------------------------------------------------------
  // Split in two so that a kernel can put every load of the board in flight BEFORE anything waits for the
  // header: issue_hdr, load_army_narrow, load_planes, land(), decode_hdr - one memory round trip per board
  // instead of two (the turn is a latency chain: a second round trip is ~15 % of a wave's life).
  __device__ __forceinline__ void issue_hdr(const uint32_t* hdr_env) {
    const int lane = lane_id();
    hv = (lane < HDR_DW) ? hdr_env[lane] : 0u;
#pragma unroll
    for (int k = 0; k < NR; ++k) rowbit[k] = 1u << lane_player(k);
  }
  __device__ __forceinline__ void decode_hdr() {
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
  // The same fields through the scalar cache (s_load: no vector instruction, no cross-lane read).  The header
  // is only written by this wave's own final store, and the scalar cache is invalidated between launches.
  __device__ __forceinline__ void decode_hdr_scalar(const uint32_t* hdr_env) {
    typedef const __attribute__((address_space(4))) uint32_t* kptr;
    kptr k = (kptr)hdr_env;
    turn = (int)k[H_TURN];
    const uint32_t dims = k[H_DIMS];
    W = (int)(dims & 0xFFu);
    H = (int)((dims >> 8) & 0xFFu);
    P = (int)((dims >> 16) & 0xFFu);
    hflags = dims >> 24;
    N = W * H;
    alive = k[H_STATUS] & 0xFFu;
    recipW = (int)k[H_RECIPW];
  }
  // keeps the scalar loads above from being sunk to their first use (a late s_load is a late round trip)
  __device__ __forceinline__ void land_scalars() {
    asm volatile("" : "+s"(turn), "+s"(W), "+s"(H), "+s"(P), "+s"(hflags), "+s"(alive), "+s"(recipW));
  }
  __device__ __forceinline__ void load_hdr(const uint32_t* hdr_env) {
    issue_hdr(hdr_env);
    decode_hdr();
  }
  // Every register a load of this board writes is "touched" here: the compiler can neither sink those loads below a
  // later (header-dependent) branch nor split the wait - all of them are in flight together and land at this point.
  // The armies are not waited for here: issued last, they are the tail of the load burst, and the turn needs them
  // only at the action phase - after the agent has sampled its moves from the planes (measured: neutral, 223.3 vs 222.7 us).
  __device__ __forceinline__ void land() {
    asm volatile("" : "+v"(hv));
#pragma unroll
    for (int k = 0; k < NR; ++k) {
      asm volatile("" : "+v"(own[k]));
      asm volatile("" : "+v"(lst[k]));
      asm volatile("" : "+v"(vis[k]));
    }
#pragma unroll
    for (int k = 0; k < NSR; ++k) asm volatile("" : "+v"(shp[k]));
  }
  // profiling builds: makes the state opaque to the optimiser between two runs of the same phase
  __device__ __forceinline__ void opaque() {
    opaque_v();
    asm volatile("" : "+s"(turn), "+s"(alive), "+s"(hflags));
  }
  __device__ __forceinline__ void opaque_v() {
    asm volatile("" : "+v"(hv));
#pragma unroll
    for (int s = 0; s < NSLOT; ++s) asm volatile("" : "+v"(army[s]));
#pragma unroll
    for (int k = 0; k < NR; ++k) {
      asm volatile("" : "+v"(own[k]));
      asm volatile("" : "+v"(lst[k]));
      asm volatile("" : "+v"(vis[k]));
    }
    asm volatile("" : "+v"(chg));
    asm volatile("" : "+v"(vch));
    asm volatile("" : "+v"(gt1));
  }
  __device__ __forceinline__ void hdr_set(int k, uint32_t v) { hv = (uint32_t)gvec_llvm_writelane((int)v, k, (int)hv); }
  __device__ __forceinline__ uint32_t hdr_get(int k) const { return rdlane(hv, k); }
  __device__ __forceinline__ void store_hdr(uint32_t* hdr_env, uint32_t last_err) {
    hdr_set(H_TURN, (uint32_t)turn);
    hdr_set(H_DIMS, (uint32_t)W | ((uint32_t)H << 8) | ((uint32_t)P << 16) | (hflags << 24));
    hdr_set(H_STATUS, alive | (last_err << 16));  // (H_RECIPW is a function of W: it rides along in hv unchanged)
    if (lane_id() < HDR_DW) st_stream<GVEC_NT_PLANE>(hdr_env + lane_id(), hv);
  }

  // ---- planes: one load per PPR planes (row r of a register = plane base + r) ----------------------------
  // Lanes that hold no dword of the bit string (column >= fd, or a row without a plane) must read as zero.
  // They are pointed at `zeros` - a block of at least row_dw zero dwords - so every load is ONE instruction with an
  // immediate offset and no select on the data.  The shared planes come in PPR at a time as well (a load per shared
  // plane with every row reading the same dwords is 13 vector-memory instructions instead of 4, and the memory
  // pipeline, not the bytes, then sets the pace: scripts/microbench/copy_pattern2.hip) and are spread over the rows
  // afterwards, one ds_bpermute each (spread_shared).  PACK = false: the shared planes one load each, every row reading
  // the same dwords - for the kernels that load a board once and then play many turns on it (no spread_shared, and
  // no registers held for it).
  template <bool PACK = true>
  __device__ __forceinline__ void load_planes(const uint32_t* rows_env, int fd, const uint32_t* zeros) {
    const bool in = col() < fd;
    const uint32_t* lane_base = rows_env + (row() * fd + col());
    if constexpr (PACK) {
      // the per-turn kernel: lanes without a dword do not load at all (every wave of the chip reading the one zero
      // block makes its cache lines a hot spot of the L2 channels they live in)
#pragma unroll
      for (int k = 0; k < NR; ++k) {
        own[k] = vis[k] = 0u;
        if (in && (k * PPR + row() < MAXP)) {
          own[k] = ld_stream(lane_base + (PL::OWN + k * PPR) * fd);
          vis[k] = ld_stream(lane_base + (PL::VIS + k * PPR) * fd);
        }
      }  // the lists: load_lists(), once the header has landed
#pragma unroll
      for (int k = 0; k < NSR; ++k) {
        shp[k] = 0u;
        if (in && (k * PPR + row() < NSHARED)) shp[k] = ld_stream(lane_base + (PL::CHG + k * PPR) * fd);
      }
    } else {
#pragma unroll
      for (int k = 0; k < NR; ++k) {
        const uint32_t* gp = (in && (k * PPR + row() < MAXP)) ? lane_base : zeros;  // packed: plane (base + row), dword col
        own[k] = ld_stream(gp + (PL::OWN + k * PPR) * fd);
        vis[k] = ld_stream(gp + (PL::VIS + k * PPR) * fd);
      }
      load_lists(rows_env, fd);  // this order has the header already
      const uint32_t* gs = in ? rows_env + col() : zeros;
      chg = ld_stream(gs + PL::CHG * fd);
      vch = ld_stream(gs + PL::VCH * fd);
      gt1 = ld_stream(gs + PL::GT1 * fd);
      gen = ld_stream(gs + PL::GEN * fd);
      city = ld_stream(gs + PL::CITY * fd);
      mtn = ld_stream(gs + PL::MTN * fd);
      valid = ld_stream(gs + PL::VALID * fd);
      ncol0 = ld_stream(gs + PL::NCOL0 * fd);
      ncolL = ld_stream(gs + PL::NCOLL * fd);
#pragma unroll
      for (int d = 0; d < 4; ++d) ok[d] = ld_stream(gs + (PL::OK + d) * fd);
    }
  }
  // OwnedTiles: the board's ownership unless the header says otherwise (HF_LDIFF: a second round trip, for the few
  // envs an aborted turn has left out of step, H6).  Needs hflags and own[].
  __device__ __forceinline__ void load_lists(const uint32_t* rows_env, int fd) {
#pragma unroll
    for (int k = 0; k < NR; ++k) lst[k] = own[k];
    if (hflags & HF_LDIFF) {
      const uint32_t* lane_base = rows_env + (row() * fd + col());
#pragma unroll
      for (int k = 0; k < NR; ++k) {
        lst[k] = 0u;
        if (col() < fd && (k * PPR + row() < MAXP)) lst[k] = lane_base[(PL::LST + k * PPR) * fd];
      }
    }
  }
  // Sets HF_LDIFF to what the lists are now: call BEFORE store_hdr; the plane stores write the lists accordingly.
  __device__ __forceinline__ void settle_lists() { hflags = lists_match() ? (hflags & ~HF_LDIFF) : (hflags | HF_LDIFF); }
  __device__ __forceinline__ void store_lists(uint32_t* rows_env, int fd) const {
    if (!(hflags & HF_LDIFF)) return;
    uint32_t* gp = rows_env + row() * fd + col();
#pragma unroll
    for (int k = 0; k < NR; ++k)
      if (col() < fd && (k * PPR + row() < MAXP)) gp[(PL::LST + k * PPR) * fd] = lst[k];
  }
  // shared plane j (CHG + j), replicated in every row
  __device__ __forceinline__ uint32_t shared_plane(int j) const { return bperm((((j % PPR) * ROWL) + col()) << 2, shp[j / PPR]); }
  __device__ __forceinline__ void spread_shared() {
    chg = shared_plane(PL::CHG - PL::CHG);
    vch = shared_plane(PL::VCH - PL::CHG);
    gt1 = shared_plane(PL::GT1 - PL::CHG);
    gen = shared_plane(PL::GEN - PL::CHG);
    city = shared_plane(PL::CITY - PL::CHG);
    mtn = shared_plane(PL::MTN - PL::CHG);
    valid = shared_plane(PL::VALID - PL::CHG);
    ncol0 = shared_plane(PL::NCOL0 - PL::CHG);
    ncolL = shared_plane(PL::NCOLL - PL::CHG);
#pragma unroll
    for (int d = 0; d < 4; ++d) ok[d] = shared_plane(PL::OK + d - PL::CHG);
  }
  // The planes from GEN on change only when the env is re-dealt (with_types).
  __device__ __forceinline__ void store_planes(uint32_t* rows_env, int fd, int row_dw, bool with_types) const {
    const bool in = col() < fd;
    uint32_t* gp = rows_env + row() * fd + col();
#pragma unroll
    for (int k = 0; k < NR; ++k) {
      if (in && (k * PPR + row() < MAXP)) {
        st_stream<GVEC_NT_PLANE>(gp + (PL::OWN + k * PPR) * fd, own[k]);
        st_stream<GVEC_NT_PLANE>(gp + (PL::VIS + k * PPR) * fd, vis[k]);
      }
    }
    store_lists(rows_env, fd);
    static_assert(PL::VCH == PL::CHG + 1 && PL::GT1 == PL::CHG + 2, "the three mutable shared planes are stored as rows of one register");
#pragma unroll
    for (int k = 0; k * PPR < 3; ++k) {
      const int j = k * PPR + row();
      if (in && j < 3) st_stream<GVEC_NT_PLANE>(gp + (PL::CHG + k * PPR) * fd, j == 0 ? chg : (j == 1 ? vch : gt1));
    }
    const int lane = lane_id();
    if (lane < fd) {  // lanes 0..fd-1 are row 0, columns 0..fd-1 (fd <= ROWL)
      uint32_t* g = rows_env + lane;
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
  // narrow (u16 pairs) / wide (int32 escape) army storage: see gvec_device.hpp "army storage"
  __device__ __forceinline__ void load_army(const ArmyCRef& a) {
    army_load_narrow<NSLOT>(army, a.n);
    if (hflags & HF_WIDE) army_load_wide<NSLOT>(army, a.w);
  }
  __device__ __forceinline__ void load_army_narrow(const ArmyCRef& a) { army_load_narrow<NSLOT>(army, a.n); }  // needs no header
  __device__ __forceinline__ void load_army_wide_if_flagged(const ArmyCRef& a) {
    if (hflags & HF_WIDE) army_load_wide<NSLOT>(army, a.w);
  }
  __device__ __forceinline__ void store_army(const ArmyRef& a) {  // sets / clears HF_WIDE: call BEFORE store_hdr
    if (army_fits_narrow<NSLOT>(army)) {
      hflags &= ~HF_WIDE;
      army_store_narrow<NSLOT>(army, a.n);
    } else {
      hflags |= HF_WIDE;
      army_store_wide<NSLOT>(army, a.w);
    }
  }

  // ---- LDS army shadow of the action phase ---------------------------------------------------------
  // During the action phase the armies live in an LDS shadow: a wave-uniform tile index is one
  // broadcast ds_read / one single-lane ds_write instead of an NSLOT-way register select chain.
  __device__ __forceinline__ void army_to_lds() {
#pragma unroll
    for (int s = 0; s < NSLOT; ++s) larmy[64 * s + lane_id()] = army[s];
    wave_lds_fence();
  }
  __device__ __forceinline__ void army_from_lds() {
    wave_lds_fence();
#pragma unroll
    for (int s = 0; s < NSLOT; ++s) army[s] = larmy[64 * s + lane_id()];
  }
  static __device__ __forceinline__ void larmy_store(int32_t* p, int32_t v) { *p = v; }
  __device__ __forceinline__ int army_get(int t) const { return uni(larmy[t]); }
  __device__ __forceinline__ void army_set(int t, int val) {
    if (lane_id() == 0) larmy[t] = val;
  }

  // ---- single tiles ------------------------------------------------------------------------------------
  __device__ __forceinline__ bool bit_at(uint32_t shared_plane, int t) const { return (rdlane(shared_plane, t >> 5) >> (t & 31)) & 1u; }
  __device__ __forceinline__ bool bit_at_p(const uint32_t (&reg)[NR], int p, int t) const {
    uint32_t w = 0u;
#pragma unroll
    for (int k = 0; k < NR; ++k) {
      const uint32_t r = rdlane(reg[k], (p % PPR) * ROWL + (t >> 5));
      w = (k == p / PPR) ? r : w;
    }
    return (w >> (t & 31)) & 1u;
  }
  // the single bit of tile t, in every row
  __device__ __forceinline__ uint32_t tile_bit(int t) const { return (col() == (t >> 5)) ? (1u << (t & 31)) : 0u; }

  // ---- flat <-> tile domain ---------------------------------------------------------------------------
  // tile 64s+l is bit l&31 of dword 2s + (l>>5); row 0 of a replicated plane, row (p % PPR) of a packed one
  __device__ __forceinline__ int32_t gather_mask(uint32_t plane, int s, int row_base = 0) const {
    const int lane = lane_id();
    return __builtin_amdgcn_sbfe((int32_t)bperm(((row_base + (lane >> 5)) << 2) + 8 * s, plane), (uint32_t)(lane & 31), 1u);
  }
  __device__ __forceinline__ uint32_t gather(uint32_t plane, int s, int row_base = 0) const {  // 0 / 1
    const int lane = lane_id();
    return __builtin_amdgcn_ubfe(bperm(((row_base + (lane >> 5)) << 2) + 8 * s, plane), (uint32_t)(lane & 31), 1u);
  }
  static __device__ __forceinline__ uint32_t replicate_row0(uint32_t plane) { return bperm(col() << 2, plane); }
  // popcount of a replicated plane: row 0's lanes summed (the scan's zero fill keeps other rows out)
  __device__ __forceinline__ int count_shared(uint32_t plane) const {
    return (int)rdlane(row_scan_add<ROWL>((uint32_t)__builtin_popcount(plane)), ROWL - 1);
  }
  static __device__ __forceinline__ bool any_bit(uint32_t plane) { return __builtin_amdgcn_ballot_w64(plane != 0u) != 0ull; }
  // OwnedTiles == board ownership for every player (no owned-but-unlisted tile, H6)
  __device__ __forceinline__ bool lists_match() const {
    uint32_t d = 0u;
#pragma unroll
    for (int k = 0; k < NR; ++k) d |= lst[k] ^ own[k];
    return !any_bit(d);
  }
  // Tile.Army > 1 as a replicated flat plane: the ballot of slot s is dwords 2s, 2s+1 of the bit string
  // (written into row 0 with v_writelane: 2*NSLOT <= ROWL), one ds_bpermute copies row 0 into every row
  __device__ __forceinline__ void refresh_gt1() {
    uint32_t g = 0u;
#pragma unroll
    for (int s = 0; s < NSLOT; ++s) {
      const unsigned long long b = __builtin_amdgcn_ballot_w64(army[s] > 1);
      g = (uint32_t)gvec_llvm_writelane((int)(uint32_t)b, 2 * s, (int)g);
      g = (uint32_t)gvec_llvm_writelane((int)(uint32_t)(b >> 32), 2 * s + 1, (int)g);
    }
    gt1 = replicate_row0(g);
  }

  // ---- Engine.updateFogOfWarOptimized (visibility_optimized.go:16-97) ---------------------------------
  // matched: lists_match() holds (the caller needs it too)
  __device__ __forceinline__ void update_fog(bool matched = false) {
    if (!(hflags & HF_FOG)) return;  // :17-19
    bool full, none;
    if ((hflags & HF_VSMALL) && turn > 0 && P <= N / 10) {  // at most P tiles in V: below the threshold whatever they are
      full = false;
      none = !any_bit(vch);
    } else {
      const int nv = count_shared(vch);
      full = turn == 0 || nv > N / 10;
      none = nv == 0;
    }
    if (full) {  // :22-26 full: clear, then 3x3 around every listed tile of alive players (:33-53)
#pragma unroll
      for (int k = 0; k < NR; ++k) {
        const uint32_t d = dil3(lst[k]);  // all players of the register at once
        vis[k] = lane_flag(alive, k) ? d : 0u;
      }
      return;
    }
    if (none) return;  // incremental update over an empty set is the identity
    // :56-97 affected = board owners within 5x5 of V (:100-116); clear all players in 3x3 of V
    // (:132-150); re-light affected, alive players from their lists (:85-94).
    // "p owns a tile within 5x5 of V" <=> the 3x3 hull of p's tiles meets the 3x3 hull of V (two steps of a
    // king's walk on the board); with lists == ownership that hull is the one the re-lighting needs anyway.
    const uint32_t near3 = dil3(vch), clr = ~near3;
    const uint32_t near5 = matched ? 0u : dil3(near3);
#pragma unroll
    for (int k = 0; k < NR; ++k) {
      const uint32_t d = dil3(lst[k]);
      // affected: some lane of my row holds an owned tile near V - the row's slice of the ballot, looked at from the
      // lanes (a per-player scalar loop costs five scalar instructions a player; the scalar unit is the busier one)
      const unsigned long long hit = __builtin_amdgcn_ballot_w64((matched ? (d & near3) : (own[k] & near5)) != 0u);
      const uint32_t mine = (uint32_t)(hit >> (row() * ROWL)) & (ROWL == 32 ? 0xFFFFFFFFu : 0xFFFFu);
      const bool relit = mine != 0u && lane_flag(alive, k);
      vis[k] = (vis[k] & clr) | (relit ? d : 0u);
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

  // ---- Engine.updatePlayerStats (stats.go:8-144) ------------------------------------------------------
  // end_of_turn: the turn's one pass, no elimination before it - ChangedTiles is then at most two tiles per player plus
  //   what production raised, which outside growth turns is a subset of the generals and cities (HF_FEWSPECIAL bounds it).
  // delta: additionally the stats were exact when the turn began (HF_SYNC), lists equalled ownership, and the moves
  //   went through act_vector (or nobody moved): ArmyCount' = ArmyCount + move_delta + rate * |produced tiles the player
  //   holds now| - what the sums over the new lists give, without visiting the armies.  Lists that did not change keep
  //   Alive / GeneralIdx as the last pass left them.
  __device__ __forceinline__ void update_stats(bool end_of_turn = false, bool delta = false) {
    bool full;
    if (end_of_turn && !grow && (hflags & HF_FEWSPECIAL) && turn > 0) {
      if (!any_bit(chg)) return;  // :10-14
      full = false;               // |C| <= 2P + generals + cities <= N/5
    } else {
      const int nc = count_shared(chg);
      if (nc == 0 && turn > 0) return;       // :10-14
      full = turn == 0 || nc > N / 5;        // :20-21
    }
    uint32_t moved = 0u;
#pragma unroll
    for (int k = 0; k < NR; ++k) {
      const uint32_t nl = full ? own[k] : (own[k] & (lst[k] | chg));  // :33-49 / :90-130
      moved |= nl ^ lst[k];
      lst[k] = nl;
    }
    const int lane = lane_id();
    if (delta && !full && prod_uniform) {
      uint32_t add = dpp0<0x114>((lane < MAXP) ? (uint32_t)move_delta : 0u);  // row_shr:4: lane p -> header lane H_ARMYCNT + p
      static_assert(H_ARMYCNT == 4 && MAXP <= 8, "row_shr:4 lands lanes 0..MAXP-1 on the ArmyCount lanes");
      if (prod_rate != 0) {
#pragma unroll
        for (int k = 0; k < NR; ++k) {
          const uint32_t sc = row_scan_add<ROWL>((uint32_t)__builtin_popcount(prod_mask & lst[k]));  // row totals in the rows' last lanes
          const int p = lane - H_ARMYCNT;
          const uint32_t got = bperm((((p & (PPR - 1)) * ROWL) + ROWL - 1) << 2, sc);
          add += (p >= 0 && p < MAXP && p / PPR == k) ? __umul24(got, (uint32_t)prod_rate) : 0u;  // rates < 2^24 (gvec_create), got <= 1024
        }
      }
      hv += (lane >= H_ARMYCNT && lane < H_ARMYCNT + MAXP) ? add : 0u;
      hflags |= HF_SYNC;
      if (!any_bit(moved)) return;  // same lists, same generals on them
    } else {
      int32_t acc[MAXP];
#pragma unroll
      for (int p = 0; p < MAXP; ++p) acc[p] = 0;
      if (small) {  // bit * army + acc in one full-rate v_mad_u32_u24
#pragma unroll
        for (int s = 0; s < NSLOT; ++s) {
#pragma unroll
          for (int p = 0; p < MAXP; ++p)
            acc[p] = (int32_t)mad24((uint32_t)army[s], gather(lst[p / PPR], s, (p % PPR) * ROWL), (uint32_t)acc[p]);
        }
      } else {
#pragma unroll
        for (int s = 0; s < NSLOT; ++s) {
#pragma unroll
          for (int p = 0; p < MAXP; ++p) acc[p] += army[s] & gather_mask(lst[p / PPR], s, (p % PPR) * ROWL);
        }
      }
      // Player.ArmyCount: header lanes H_ARMYCNT .. H_ARMYCNT+MAXP-1
      const uint32_t tot = multi_sum(acc);
      hv = (lane >= H_ARMYCNT && lane < H_ARMYCNT + MAXP) ? tot : hv;
      hflags |= HF_SYNC;
    }
    // GeneralIdx / Alive (:46,52-54 / :101,122,133-135).  The reference keeps the last general in list order; with two or
    // more generals that order depends on Go map iteration.  Here: the highest listed general tile - found on lanes:
    // every lane's candidate, the row's maximum by DPP, one ds_bpermute hands row p's result to header lane H_GIDX + p.
    {
      const int hl = lane - H_GIDX;  // the player whose GeneralIdx this header lane holds
#pragma unroll
      for (int k = 0; k < NR; ++k) {
        const uint32_t g = lst[k] & gen;
        int32_t v = (g != 0u) ? (32 * col() + 31 - __builtin_clz(g)) : -1;
        v = max(v, (int32_t)__builtin_amdgcn_update_dpp(-1, v, 0x111, 0xf, 0xf, false));  // row_shr:1, lanes without a source keep -1
        v = max(v, (int32_t)__builtin_amdgcn_update_dpp(-1, v, 0x112, 0xf, 0xf, false));
        v = max(v, (int32_t)__builtin_amdgcn_update_dpp(-1, v, 0x114, 0xf, 0xf, false));
        v = max(v, (int32_t)__builtin_amdgcn_update_dpp(-1, v, 0x118, 0xf, 0xf, false));
        if constexpr (ROWL == 32) v = max(v, (int32_t)__builtin_amdgcn_update_dpp(-1, v, 0x142, 0xa, 0xf, false));  // row_bcast15 into DPP rows 1, 3
        const uint32_t got = bperm((((hl & (PPR - 1)) * ROWL) + ROWL - 1) << 2, (uint32_t)v);
        hv = (hl >= 0 && hl < MAXP && hl / PPR == k) ? got : hv;
      }
      alive = (uint32_t)(__builtin_amdgcn_ballot_w64(hl >= 0 && hl < MAXP && (int32_t)hv >= 0) >> H_GIDX) & ((1u << MAXP) - 1u);
    }
  }

  // ---- ProductionManager.ProcessTurnProduction (production_manager.go:26-101) ------------------------
  // interval_magic = ceil(2^32 / interval) (host): turn % interval without the vector unit's float reciprocal
  // (a scalar `%` by a run-time value compiles to v_cvt / v_rcp / v_mul / v_cvt + 20 scalar instructions).
  // q = mulhi(turn, magic) is floor(turn / interval) or one more while turn * interval < 2^32; beyond that the plain `%`.
  __device__ __forceinline__ void production(int pg, int pc, int pn, int interval, uint32_t interval_magic) {
    if ((uint32_t)turn < 0x10000u && (uint32_t)interval < 0x10000u) {
      const uint32_t q = __umulhi((uint32_t)turn, interval_magic);  // wave-uniform operands: s_mul_hi_u32
      int32_t r = turn - (int32_t)(q * (uint32_t)interval);
      r = r < 0 ? r + interval : r;
      grow = r == 0;
    } else {
      grow = (turn % interval) == 0;  // :27
    }
    uint32_t listed_alive = 0u;                // :39-45: lists of alive players, owner NOT re-checked (H7)
#pragma unroll
    for (int k = 0; k < NR; ++k) listed_alive |= lane_flag(alive, k) ? lst[k] : 0u;
    listed_alive = or_rows(listed_alive);
    const uint32_t mg = (pg > 0) ? (listed_alive & gen) : 0u;
    const uint32_t mc = (pc > 0) ? (listed_alive & city) : 0u;
    uint32_t mn = 0u;
    if (grow && pn > 0) mn = listed_alive & ~(gen | city | mtn) & valid;  // wave-uniform: one turn in `interval`
    chg |= mg | mc | mn;  // :59-61 (prod > 0 only)
    const int an = (grow && pn > 0) ? pn : pc;
    prod_uniform = pg == pc && pc == an;
    prod_mask = mg | mc | mn;
    prod_rate = pg;
    if (prod_uniform) {  // one rate for every producing tile: one gather per slot
      const uint32_t m = prod_mask;
#pragma unroll
      for (int s = 0; s < NSLOT; ++s) army[s] = (int32_t)mad24(gather(m, s), (uint32_t)pg, (uint32_t)army[s]);  // rates < 2^24 (gvec_create)
    } else {
#pragma unroll
      for (int s = 0; s < NSLOT; ++s) army[s] += (pg & gather_mask(mg, s)) + (pc & gather_mask(mc, s)) + (pn & gather_mask(mn, s));
    }
  }

  // ---- WinConditionChecker.CheckGameOver (rules/win_conditions.go:21-57) ---------------------------
  __device__ __forceinline__ void check_game_over() {
    const int na = __builtin_popcount(alive);
    const int most = (P > 1) ? 1 : 0;  // originalPlayers == len(Players): over when at most one (none, for a 1-player board) is alive
    const bool over = na <= most;
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
      if (!bit_at_p(own, PID, ft)) code = GVEC_ERR_NOT_OWNED;         // action.go:82-84
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
    if (bit_at_p(own, PID, tt)) {  // :62-66 own tile: consolidate
      army_set(tt, ta + n);
    } else if (n > ta) {  // :69-82 capture (ties favour the defender)
      int prev = -1;
#pragma unroll
      for (int q = 0; q < MAXP; ++q)
        if (bit_at_p(own, q, tt)) prev = q;
#pragma unroll
      for (int k = 0; k < NR; ++k) {
        own[k] &= ~tbit;  // the bit sits in every row: cleared for every player
        if (k == PID / PPR) own[k] |= in_row_of(PID) ? tbit : 0u;
      }
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
  // ---- the action phase, all players at once ---------------------------------------------------------------
  // ActionProcessor.ProcessActions applies the moves in PlayerID order, each against the board the lower ids left
  // behind.  When no tile is touched by two of this turn's moves - the overwhelmingly common case - every move only sees
  // pre-turn state, and the whole phase runs once on lanes (lane p = player p) instead of P times on the scalar unit:
  // Validate's state-dependent half (core/action.go:82-98), ApplyMoveAction (core/movement.go:23-89) and the captures'
  // bookkeeping.  Tile coincidence is detected exactly (every candidate move ORs its two tile bits into an LDS bitmap:
  // all distinct iff the bitmap's popcount is twice the number of candidates); otherwise - returns false, nothing
  // modified - the caller takes the sequential path below.  Needs the armies in the LDS shadow.
  __device__ __forceinline__ bool act_vector(const ActVec& av, uint32_t& first_err, uint64_t& orders, int& n_orders, uint32_t& elim_seen) {
    const int lane = lane_id();
    uint32_t* s_cand = lscr;
    uint32_t* s_chg = lscr + ROWL;
    uint32_t* s_cap = lscr + 2 * ROWL;
    uint32_t* s_own = lscr + 3 * ROWL;
    uint32_t* s_loss = lscr + 3 * ROWL + 64 * NR;
    uint32_t* s_dummy = lscr + ACT_DUMMY + lane;  // per lane: where an inactive lane's LDS writes land
#pragma unroll
    for (int i = 0; i < ACT_SCRATCH_DW / 64; ++i) lscr[64 * i + lane] = 0u;
    const uint32_t m = av.meta;
    const bool active = lane < P && (m & 16u) != 0u && ((alive >> lane) & 1u) != 0u;  // action_processor.go:56-60 (H2)
    const uint32_t scode = m & 15u;
    const bool cand = active && scode == 0u;
    const int ft = cand ? av.ft : 0, tt = cand ? av.tt : 0;
    const uint32_t fbit = 1u << (ft & 31), tbit = 1u << (tt & 31);
    const int fcol = ft >> 5, tcol = tt >> 5;
    wave_lds_fence();
    atomicOr(cand ? &s_cand[fcol] : s_dummy, fbit);
    atomicOr(cand ? &s_cand[tcol] : s_dummy, tbit);
    wave_lds_fence();
    // every row sums the same ROWL dwords: no lane needs masking, row 0's last lane is read
    const int distinct = (int)rdlane(row_scan_add<ROWL>((uint32_t)__builtin_popcount(s_cand[lane & (ROWL - 1)])), ROWL - 1);
    const int ncand = __builtin_popcountll(__builtin_amdgcn_ballot_w64(cand));
    if (distinct != 2 * ncand) return false;  // two moves meet on a tile: order matters
    // pre-turn facts of my move
    const int32_t fa = larmy[ft], ta = larmy[tt];
    const int myrow = (lane % PPR) * ROWL;
    uint32_t w_ft = 0u;  // my own ownership row at the source dword
#pragma unroll
    for (int k = 0; k < NR; ++k) {
      const uint32_t g = bperm((myrow + fcol) << 2, own[k]);
      w_ft = (lane / PPR == k) ? g : w_ft;
    }
    uint32_t owner_bits = 0u;  // bit q: player q owns my target tile (pre-turn; none: neutral)
#pragma unroll
    for (int q = 0; q < MAXP; ++q) {
      const uint32_t g = bperm((((q % PPR) * ROWL) + tcol) << 2, own[q / PPR]);
      owner_bits |= ((g >> (tt & 31)) & 1u) << q;
    }
    const bool own_ft = ((w_ft >> (ft & 31)) & 1u) != 0u;
    const bool mine_tt = ((owner_bits >> lane) & 1u) != 0u;  // lane < MAXP wherever it matters (cand)
    const bool mtn_tt = ((bperm(tcol << 2, mtn) >> (tt & 31)) & 1u) != 0u;
    const bool gen_tt = ((bperm(tcol << 2, gen) >> (tt & 31)) & 1u) != 0u;
    // MoveAction.Validate, the state-dependent half, in the reference's order (action.go:82-98): the last assignment
    // that applies is the first check that fails
    uint32_t dyn = mtn_tt ? GVEC_ERR_TARGET_IS_MOUNTAIN : 0u;
    dyn = (fa <= 1) ? GVEC_ERR_INSUFFICIENT_ARMY : dyn;
    dyn = own_ft ? dyn : GVEC_ERR_NOT_OWNED;
    const uint32_t code = cand ? dyn : scode;
    const bool ok = cand && dyn == 0u;
    {  // the FIRST error in PlayerID order (action_processor.go:66-77)
      const unsigned long long em = __builtin_amdgcn_ballot_w64(active && code != 0u);
      if (em && !first_err) first_err = rdlane(code, (int)__builtin_ctzll(em));
    }
    // core.ApplyMoveAction (movement.go:40-86)
    int32_t n = (m & 32u) ? (fa >> 1) : (fa - 1);  // fa >= 2 where it matters
    n = n < 1 ? 1 : n;
    const bool take = !mine_tt && n > ta;  // ties favour the defender
    const bool capture = ok && take;
    const int32_t fight = take ? n - ta : ta - n;
    const int32_t new_ta = mine_tt ? ta + n : fight;
    larmy_store(ok ? &larmy[ft] : reinterpret_cast<int32_t*>(s_dummy), fa - n);
    larmy_store(ok ? &larmy[tt] : reinterpret_cast<int32_t*>(s_dummy), new_ta);
    atomicOr(ok ? &s_chg[fcol] : s_dummy, fbit);  // :57-60
    atomicOr(ok ? &s_chg[tcol] : s_dummy, tbit);
    atomicOr(capture ? &s_cap[tcol] : s_dummy, tbit);                              // action_processor.go:84-86
    atomicOr(capture ? &s_own[64 * (lane / PPR) + myrow + tcol] : s_dummy, tbit);  // :69-82 the tile is mine now
    // What the move does to the armies on LISTED tiles when lists equal ownership (update_stats' delta path): on my own
    // tile nothing moves out of my hands; anywhere else the fight burns `loss` on both sides - mine (-n at the source,
    // + n - ta at a captured target) and the previous owner's (the tile's ta with the tile, or n off its army).
    const int32_t burnt = take ? ta : n;
    const int32_t loss = (ok && !mine_tt) ? burnt : 0;
    atomicAdd((loss != 0 && owner_bits != 0u) ? &s_loss[__builtin_ctz(owner_bits | 0x80000000u) & 7] : s_dummy, (uint32_t)loss);
    wave_lds_fence();
    move_delta = -(loss + (int32_t)s_loss[lane & 7]);
    const uint32_t capbits = s_cap[col()];
    chg |= s_chg[col()];
    vch |= capbits;
#pragma unroll
    for (int k = 0; k < NR; ++k) own[k] = (own[k] & ~capbits) | s_own[64 * k + lane];
    // core.ProcessCaptures (movement.go:100-118): captured generals with a previous owner, in PlayerID order
    unsigned long long el = __builtin_amdgcn_ballot_w64(capture && gen_tt);
    if (el) el = __builtin_amdgcn_ballot_w64(capture && gen_tt && owner_bits != 0u);  // a neutral general: no order
    while (el) {
      const int p = (int)__builtin_ctzll(el);
      el &= el - 1ull;
      const int prev = (int)__builtin_ctz(rdlane(owner_bits, p));
      if (!((elim_seen >> prev) & 1u)) {
        orders |= (uint64_t)((uint32_t)prev | ((uint32_t)p << 4)) << (8 * n_orders);
        n_orders++;
        elim_seen |= 1u << prev;
      }
    }
    return true;
  }

  template <int PID>
  __device__ __forceinline__ void act_chain(const ActVec& av, uint32_t& first_err, uint64_t& orders, int& n_orders,
                                            uint32_t& elim_seen) {
    if constexpr (PID < MAXP) {
      if (PID < P) apply_action<PID>(av, first_err, orders, n_orders, elim_seen);
      act_chain<PID + 1>(av, first_err, orders, n_orders, elim_seen);
    }
  }

  // ---- Engine.handleEliminationsAndTileTurnover (engine.go:118-152) ----------------------------------
  __device__ __forceinline__ void eliminate(uint64_t orders, int n_orders) {
    for (int e = 0; e < n_orders; ++e) {
      const int v = (int)((orders >> (8 * e)) & 15u), nw = (int)((orders >> (8 * e + 4)) & 15u);
      uint32_t tiles = 0u;  // victim's listed tiles still owned by the victim (:130-131, H4): its row only
#pragma unroll
      for (int k = 0; k < NR; ++k) tiles |= (lane_player(k) == v) ? (lst[k] & own[k]) : 0u;
      tiles = or_rows(tiles);  // now in every row
#pragma unroll
      for (int k = 0; k < NR; ++k) {
        own[k] = (lane_player(k) == v) ? (own[k] & ~tiles) : own[k];
        own[k] = (lane_player(k) == nw) ? (own[k] | tiles) : own[k];
      }
      hdr_set(H_GIDX + v, 0xFFFFFFFFu);  // :141 GeneralIdx = -1
      chg |= tiles;                      // :133-134
      vch |= tiles;
      hflags &= ~HF_VSMALL;
      alive &= ~(1u << v);  // :140
    }
  }

  // ---- TurnProcessor.ProcessTurn (turn_processor.go:29-77) ------------------------------------------
  // acts_lo/hi: lane p holds player p's gvec_action words.  Returns the per-env error code.
  // Precondition: the caller has checked Engine.gameOver (validateGameState :95-113).
  __device__ __forceinline__ uint32_t turn_step(uint32_t acts_lo, uint32_t acts_hi, const StepArgs& A, bool& aborted) {
    return turn_step(prevalidate(acts_lo, acts_hi), A, aborted);
  }
  // av: lane p holds player p's move after the static checks (prevalidate, or the on-device agent's own output)
  __device__ __forceinline__ uint32_t turn_step(const ActVec& av, const StepArgs& A, bool& aborted) {
    aborted = false;
    turn++;  // initializeTurn :124-135
    const bool matched = lists_match();
    if (!(GVEC_PROFILE_SKIP & 2)) update_fog(matched);
    if (GVEC_PROFILE_DUP & 2) { opaque(); update_fog(matched); }
    chg = 0u;
    vch = 0u;
    hflags |= HF_VSMALL;  // the moves add one captured tile per player at most; a turnover clears the flag
    bool delta = matched && (hflags & HF_SYNC) != 0u;
    move_delta = 0;
    uint32_t first_err = 0u, elim_seen = 0u;
    uint64_t orders = 0ull;
    int n_orders = 0;
    // Engine.processActions (engine.go:80-115): PlayerID order == slot order (sort.Slice :39-41)
    const unsigned long long present = __builtin_amdgcn_ballot_w64((av.meta & 16u) != 0u && lane_id() < P);
    if (present && !(GVEC_PROFILE_SKIP & 4)) {  // a turn where nobody moves touches no army
      army_to_lds();
      if (!(GVEC_ACT_VECTOR && lscr && act_vector(av, first_err, orders, n_orders, elim_seen))) {
        act_chain<0>(av, first_err, orders, n_orders, elim_seen);
        delta = false;  // the sequential path keeps no account of the armies it moved
      }
      army_from_lds();
    }
    if (n_orders > 0) {  // engine.go:101-109
      eliminate(orders, n_orders);
      update_stats();
    }
    if (first_err) {  // engine.go:111-113 -> turn_processor.go:55-57: production, stats, game-over skipped (H5)
      if (n_orders == 0) hflags &= ~HF_SYNC;  // armies moved and no pass followed
      aborted = true;
      return first_err;
    }
    if (!(GVEC_PROFILE_SKIP & 8)) production(A.prod_general, A.prod_city, A.prod_normal, A.interval, A.interval_magic);  // :60
    if (!(GVEC_PROFILE_SKIP & 16)) update_stats(n_orders == 0, delta && n_orders == 0);                  // :65,170-179
    if (GVEC_PROFILE_DUP & 16) { opaque_v(); update_stats(n_orders == 0, false); }
    check_game_over();
    return 0u;
  }

  // ---- EngineInitializer.performInitialSetup (engine_initializer.go:218-225) -----------------
  __device__ __forceinline__ void initial_setup() {
    turn = 0;
    chg = 0u;
    vch = 0u;
#pragma unroll
    for (int k = 0; k < NR; ++k) vis[k] = 0u;
    hflags = (hflags & ~HF_DONE) | HF_VSMALL;
    update_stats();  // Turn == 0 => full
    update_fog();    // Turn == 0 => full
    check_game_over();
  }

  // ---- LegalMoveCalculator.GetLegalActionMask (rules/legal_moves.go:19-73) --------------------
  // A player's packed mask is FOUR DIRECTION BIT-PLANES of fd dwords each: bit t of plane d = action
  // (y*W+x)*4 + d of the reference, t = y*W + x.  In the flat domain plane d is
  //     (listed & owned & army > 1)  &  ok[d]     ("the d-neighbour is on the board, not a mountain")
  // for all the players of a register at once: m[k][d], row r = player k*PPR + r.  Needs a current gt1.
  // SERIALIZER = false: Engine.GetLegalActionMask, d = 0 up, 1 right, 2 down, 3 left (H10).
  // SERIALIZER = true : Serializer.GenerateActionMask (internal/experience/serializer.go:112-176):
  //   board owner (not the list), army >= 2, no Alive check, d = 0 up, 1 DOWN, 2 LEFT, 3 right (H10).
  template <bool SERIALIZER = false>
  __device__ __forceinline__ void legal_planes(uint32_t (&m)[NR][4]) const {
#pragma unroll
    for (int k = 0; k < NR; ++k) {
      // legal_moves.go :26-28 alive, :37 listed, :41 owner == pid && army > 1
      const uint32_t s = SERIALIZER ? (own[k] & gt1) : (lst[k] & own[k] & gt1);
      const uint32_t src = (SERIALIZER || lane_flag(alive, k)) ? s : 0u;
      m[k][0] = src & ok[0];
      m[k][1] = src & (SERIALIZER ? ok[2] : ok[1]);
      m[k][2] = src & (SERIALIZER ? ok[3] : ok[2]);
      m[k][3] = src & (SERIALIZER ? ok[1] : ok[3]);
    }
  }
  // legal_env: this env's [pstride][4*fd] dwords.  Lane (row r, column c) owns dword d*fd + c of player
  // k*PPR + r: every store instruction writes one direction plane of every player of the register.
  __device__ __forceinline__ void store_masks(const uint32_t (&m)[NR][4], uint32_t* legal_env, int fd, int pstride) const {
    const bool in = col() < fd;
#pragma unroll
    for (int k = 0; k < NR; ++k) {
      const int p = lane_player(k);
      if (in && p < pstride && p < MAXP) {
        uint32_t* g = legal_env + (size_t)p * (4 * fd) + col();
#pragma unroll
        for (int d = 0; d < 4; ++d) st_stream<GVEC_NT_MASK>(g + d * fd, m[k][d]);
      }
    }
  }
  // ---- stores of the per-turn kernel: through the LDS army shadow, as whole 16-byte chunks of consecutive lanes ----
  // The memory pipeline, not the bytes, sets the step kernel's pace (scripts/microbench/copy_pattern2.hip): a block
  // that leaves as ONE wide store instruction is cheaper than the same bytes in four narrow ones.  The shadow is idle
  // once the action phase is over; every routine fences before it overwrites what the previous one staged.
  template <int CLASS>
  __device__ __forceinline__ void flush_stage(uint32_t* dst, int dwords) const {  // dst 16-byte aligned
    const u32x4* s4 = reinterpret_cast<const u32x4*>(larmy);
    u32x4* g4 = reinterpret_cast<u32x4*>(dst);
    const int chunks = dwords >> 2;
    for (int i = lane_id(); i < chunks; i += 64) {
      st_stream<CLASS>(g4 + i, s4[i]);
    }
    const int tail = dwords & 3, l = lane_id();
    if (l < tail) st_stream<CLASS>(dst + 4 * chunks + l, reinterpret_cast<const uint32_t*>(larmy)[4 * chunks + l]);
  }
  // the planes that change in a turn (OWN .. GT1: contiguous in the block)
  __device__ __forceinline__ void store_planes_staged(uint32_t* rows_env, int fd) const {
    uint32_t* stage = reinterpret_cast<uint32_t*>(larmy);
    const bool in = col() < fd;
    uint32_t* g = stage + row() * fd + col();
    wave_lds_fence();
#pragma unroll
    for (int k = 0; k < NR; ++k) {
      if (in && (k * PPR + row() < MAXP)) {
        g[(PL::OWN + k * PPR) * fd] = own[k];
        g[(PL::VIS + k * PPR) * fd] = vis[k];
      }
    }
#pragma unroll
    for (int k = 0; k * PPR < 3; ++k) {
      const int j = k * PPR + row();
      if (in && j < 3) g[(PL::CHG + k * PPR) * fd] = j == 0 ? chg : (j == 1 ? vch : gt1);
    }
    wave_lds_fence();
    flush_stage<GVEC_NT_PLANE>(rows_env, PL::MUTABLE * fd);
    store_lists(rows_env, fd);
  }
  // narrow armies (the caller has checked army_fits_narrow): the block's layout is army_store_narrow's
  __device__ __forceinline__ void store_army_narrow_staged(uint32_t* n) const {
    uint32_t* stage = reinterpret_cast<uint32_t*>(larmy);
    const int lane = lane_id();
    wave_lds_fence();
#pragma unroll
    for (int k = 0; k < NSLOT / 2; ++k) stage[64 * k + lane] = (uint32_t)army[2 * k] | ((uint32_t)army[2 * k + 1] << 16);
    if constexpr ((NSLOT & 1) != 0) reinterpret_cast<uint16_t*>(stage + 64 * (NSLOT / 2))[lane] = (uint16_t)army[NSLOT - 1];
    wave_lds_fence();
    flush_stage<GVEC_NT_ARMY>(n, NSLOT * 32);
  }
  __device__ __forceinline__ void store_army_staged(const ArmyRef& a) {  // sets / clears HF_WIDE: call BEFORE store_hdr
    if (army_fits_narrow<NSLOT>(army)) {
      hflags &= ~HF_WIDE;
      store_army_narrow_staged(a.n);
    } else {
      hflags |= HF_WIDE;
      army_store_wide<NSLOT>(army, a.w);
    }
  }
  // The same bytes through the LDS army shadow (idle once the action phase is over): store_masks' instructions each
  // write one direction of every player - pieces of fd dwords, 4*fd dwords apart - and partial-line writes from four
  // instructions cost the memory pipeline 11 % of the whole step (scripts/microbench/copy_pattern2.hip).  Staged, the
  // env's mask block leaves as whole 16-byte chunks of consecutive lanes: one store instruction at 20x20 4P.
  // Needs larmy (NSLOT*64 dwords >= pstride*4*fd for MAXP <= 8); legal_env is 16-byte aligned (4*fd dwords per player).
  __device__ __forceinline__ void store_masks_staged(const uint32_t (&m)[NR][4], uint32_t* legal_env, int fd, int pstride) const {
    static_assert(MAXP <= 8, "the army shadow holds the mask block");
    if ((reinterpret_cast<uintptr_t>(legal_env) & 15u) != 0u) {  // a caller's own mask buffer (gvec_step, device pointers) may sit anywhere
      store_masks(m, legal_env, fd, pstride);
      return;
    }
    uint32_t* stage = reinterpret_cast<uint32_t*>(larmy);
    const bool in = col() < fd;
    wave_lds_fence();
#pragma unroll
    for (int k = 0; k < NR; ++k) {
      const int p = lane_player(k);
      if (in && p < pstride && p < MAXP) {
        uint32_t* g = stage + p * (4 * fd) + col();
#pragma unroll
        for (int d = 0; d < 4; ++d) g[d * fd] = m[k][d];
      }
    }
    wave_lds_fence();
    flush_stage<GVEC_NT_MASK>(legal_env, pstride * 4 * fd);
  }
};

}  // namespace gvec
