// gvec_kernels.hip — HIP kernels for gfx950 (MI355X).  One wavefront per board; see
// gvec_device.hpp for the register layout and the reference citations.
#include "gvec_launch.hpp"
#include "gvec_packed.hpp"

#include <type_traits>

namespace gvec {

constexpr int WAVES_PER_BLOCK = 4;

// All turn logic runs on PBoard (players packed into register rows, gvec_packed.hpp); Board is the plain
// layout of the conversion / experience kernels.
template <int MAXP, int NSLOT>
using Turn = PBoard<MAXP, NSLOT>;

// =========================================================================================
// random agent (DESIGN.md "Synthetic inputs"; mirrored by the oracle's agent_env)
// =========================================================================================
// Every alive player draws two hashes h1 = amix(key + turn*c1 + player*c2 + c3), h2 = amix(h1 ^ c4):
//   no action            if (h1 & 0xFFFF) < agent_noop
//   half move            if (h1 >> 16) < agent_half
//   unchecked move       if invalid_permille > 0 and ((h2 & 0xFFFF) * 1000 >> 16) < invalid_permille:
//                        tile ((h2 >> 16) * N) >> 16, direction (h1 >> 8) & 3  (H5 stress)
//   else the kk-th legal move, kk = ((h2 >> 16) * count) >> 16, of Engine.GetLegalActionMask(player) in the
//   order (t >> 5, d, t & 31): 32-tile blocks ascending, inside a block direction plane by direction plane
//   (up, right, down, left), inside a plane tiles ascending.  No legal move: no action.
// All players of a register are sampled at once: lane (row r, column c) counts the legal moves of player
// r in tile block c, one row-wise prefix scan finds each row's lane, that lane finds its bit.
// Returns, in lane p, player p's draw: t | d << 10 | act << 12 | half << 13.
template <int MAXP, int NSLOT>
__device__ __forceinline__ uint32_t agent_sample(const Turn<MAXP, NSLOT>& b, const uint32_t (&m)[Turn<MAXP, NSLOT>::NR][4], uint32_t ek,
                                                 const StepArgs& A) {
  using T = Turn<MAXP, NSLOT>;
  constexpr int NR = T::NR, PPR = T::PPR, ROWL = T::ROWL;
  const int lane = lane_id();
  const uint32_t sbase = ek + (uint32_t)b.turn * 0x9E3779B1u + 0x165667B1u;  // wave-uniform
  uint32_t mine = 0u;
#pragma unroll
  for (int k = 0; k < NR; ++k) {
    const uint32_t player = (uint32_t)T::lane_player(k);
    const uint32_t h1 = amix(mad24(player, 0x4A7C15u, sbase));
    const uint32_t h2 = amix(h1 ^ 0x68E31DA4u);
    const uint32_t hi16 = h2 >> 16;
    const bool act = b.lane_flag(b.alive, k) && !((h1 & 0xFFFFu) < A.agent_noop);  // alive implies player < P
    const bool half = (h1 >> 16) < A.agent_half;
    const bool inv = A.invalid_permille > 0 && (__umul24(h2 & 0xFFFFu, 1000u) >> 16) < (uint32_t)A.invalid_permille;
    const uint32_t c0 = (uint32_t)__builtin_popcount(m[k][0]), c1 = c0 + (uint32_t)__builtin_popcount(m[k][1]);
    const uint32_t c2 = c1 + (uint32_t)__builtin_popcount(m[k][2]), cnt = c2 + (uint32_t)__builtin_popcount(m[k][3]);
    const uint32_t sc = row_scan_add<ROWL>(cnt);
    const uint32_t total = row_last<ROWL>(sc);     // this row's number of legal moves (< 4096)
    const uint32_t kk = __umul24(hi16, total) >> 16;
    const uint32_t below = sc - cnt;
    const bool sel = kk >= below && kk < sc;       // exactly one lane of a row with total > 0
    const uint32_t r = kk - below;
    const uint32_t d = (r >= c0 ? 1u : 0u) + (r >= c1 ? 1u : 0u) + (r >= c2 ? 1u : 0u);
    const uint32_t base = (d == 0u) ? 0u : (d == 1u) ? c0 : (d == 2u) ? c1 : c2;
    const uint32_t w = (d == 0u) ? m[k][0] : (d == 1u) ? m[k][1] : (d == 2u) ? m[k][2] : m[k][3];
    const uint32_t t_sel = 32u * (uint32_t)T::col() + kth_set_bit(w, r - base);
    const uint32_t pick = row_scan_or<ROWL>(sel ? (t_sel | (d << 10) | 0x8000u) : 0u);  // complete at the row's last lane
    const uint32_t unchecked = (__umul24(hi16, (uint32_t)b.N) >> 16) | (((h1 >> 8) & 3u) << 10) | 0x8000u;
    const uint32_t fin = inv ? unchecked : pick;
    const bool go = act && (fin & 0x8000u) != 0u;
    const uint32_t out = (fin & 0xFFFu) | (go ? 0x1000u : 0u) | (half ? 0x2000u : 0u);
    // lane p <- the last lane of row p % PPR of register p / PPR
    const uint32_t got = bperm((((lane % PPR) * ROWL) + ROWL - 1) << 2, out);
    mine = (lane / PPR == k) ? got : mine;
  }
  return mine;
}

// the draw as gvec_action words (gvec_agent_actions, actions_out)
template <int MAXP, int NSLOT>
__device__ __forceinline__ void agent_words(const Turn<MAXP, NSLOT>& b, uint32_t mine, uint32_t& alo, uint32_t& ahi) {
  const bool act = lane_id() < MAXP && (mine & 0x1000u) != 0u;
  const int t = (int)(mine & 0x3FFu), d = (int)((mine >> 10) & 3u);
  const int y = (int)(__umul24((uint32_t)t, (uint32_t)b.recipW) >> 16), x = t - (int)__umul24((uint32_t)y, (uint32_t)b.W);  // t < 1024
  const int dx = (d == 1) - (d == 3), dy = (d == 2) - (d == 0);
  const uint32_t lo = ((uint32_t)x & 0xFFu) | (((uint32_t)y & 0xFFu) << 8) | (((uint32_t)(x + dx) & 0xFFu) << 16) |
                      (((uint32_t)(y + dy) & 0xFFu) << 24);
  alo = act ? lo : 0u;
  ahi = act ? (GVEC_ACT_VALID | ((mine & 0x2000u) ? GVEC_ACT_HALF : 0u)) : 0u;
}

// the draw as the turn's ActVec, skipping the coordinate round trip through gvec_action: what
// PBoard::prevalidate would derive from agent_words' output.  A legal move needs no static check (its
// target is on the board by construction of the mask); an unchecked one (invalid_permille) can only leave
// the board (core/action.go:58-64) - same tile and adjacency hold for every (tile, direction) pair.
template <int MAXP, int NSLOT>
__device__ __forceinline__ typename Turn<MAXP, NSLOT>::ActVec agent_actvec(const Turn<MAXP, NSLOT>& b, uint32_t mine, bool may_be_unchecked) {
  typename Turn<MAXP, NSLOT>::ActVec v;
  const bool act = lane_id() < MAXP && (mine & 0x1000u) != 0u;
  const int t = (int)(mine & 0x3FFu), d = (int)((mine >> 10) & 3u);
  uint32_t code = 0u;
  if (may_be_unchecked) {  // wave-uniform
    const int y = (int)(__umul24((uint32_t)t, (uint32_t)b.recipW) >> 16), x = t - (int)__umul24((uint32_t)y, (uint32_t)b.W);
    const bool off = (d == 0) ? (y == 0) : (d == 1) ? (x == b.W - 1) : (d == 2) ? (y == b.H - 1) : (x == 0);
    code = off ? GVEC_ERR_INVALID_COORDINATES : 0u;
  }
  v.meta = act ? (code | 16u | ((mine & 0x2000u) ? 32u : 0u)) : 0u;
  v.ft = t;
  v.tt = t + ((d == 0) ? -b.W : (d == 1) ? 1 : (d == 2) ? b.W : -1);
  return v;
}

// =========================================================================================
// step / rollout kernel: `turns` engine turns per launch for one board per wavefront
// =========================================================================================
template <typename BT>
__device__ __forceinline__ void load_board(BT& b, const uint32_t* hdr, const uint32_t* rows, const ArmyCRef& army, int fd) {
  b.load_hdr(hdr);
  b.load_army(army);
  b.load_planes(rows, fd);
}
// EARLY (the per-turn step kernel): every load of the board goes out before the header is decoded - one memory round
// trip per board instead of two, worth 10 % there (one-process A/B: 338.7 -> 305.6 us per 262,144 boards).  The fused
// rollout amortises its loads over many turns and runs 4 % faster with the plain order (fewer live registers).
template <bool EARLY = false, typename BT>
__device__ __forceinline__ void load_turn(BT& b, const uint32_t* hdr, const uint32_t* rows, const ArmyCRef& army, int fd, const uint32_t* zeros) {
  if constexpr (EARLY) {
    b.issue_hdr(hdr);
    b.load_planes(rows, fd, zeros);
    b.load_army_narrow(army);
    b.decode_hdr_scalar(hdr);   // SMEM: in flight with the vector loads above
    b.land();
    b.land_scalars();
    b.spread_shared();
    b.load_lists(rows, fd);
    b.load_army_wide_if_flagged(army);
  } else {
    b.load_hdr(hdr);
    b.load_army(army);
    b.template load_planes<false>(rows, fd, zeros);
  }
}

// vector-env auto-reset: this step re-deals the env from the board pool (no Go analogue)
template <int MAXP, int NSLOT, typename BT>
__device__ __forceinline__ void redeal(BT& b, const StepArgs& A, int env, int fd, int row_dw) {
  const uint32_t episode = b.hdr_get(H_EPISODE) + 1u;
  const uint32_t cs = b.hdr_get(H_CNT_STEPS), ca = b.hdr_get(H_CNT_ABORT), cd = b.hdr_get(H_CNT_DONE);
  const uint32_t hk = fmix32(env_key_of(A.pool_seed_base, (uint32_t)env) ^ (episode * 0x9E3779B1u));
  const int j = (int)__umulhi(hk, (uint32_t)A.pool_size);
  load_turn(b, A.pool_hdr + (size_t)j * HDR_DW, A.pool_rows + (size_t)j * row_dw, army_cref<NSLOT>(A.pool_army16, A.pool_army32, j), fd, A.zeros);
  b.hdr_set(H_EPISODE, episode);
  b.hdr_set(H_CNT_STEPS, cs);
  b.hdr_set(H_CNT_ABORT, ca);
  b.hdr_set(H_CNT_DONE, cd);
}

// ONE engine turn per launch, straight-line (gvec_step; per-turn rollouts).  AGENT: actions are
// sampled on device from the legal-move planes of the resident state.
// Waves per SIMD asked of the register allocator: the board state a variant holds (planes, army
// slots, mask planes) plus ~32 working registers.  <4,7> fits 64 registers = 8 waves/SIMD without a
// spill, which is worth 7 % over 7 waves (one-process A/B): the turn is a long dependent chain of
// short cross-lane operations, and the VALU only stays fed with every wave slot occupied.
constexpr int step_waves(int maxp, int nslot) {
  const int ppr = (nslot <= 7) ? 4 : 2;  // PBoard: players per plane register
  const int nr = (maxp + ppr - 1) / ppr;
  const int state = 3 * nr + 13 + nr + nslot + 4 * nr;  // packed planes, shared planes, rowbit, armies, mask planes
  // 8 waves only where they fit with room to spare
  const int need = state + (state + 32 <= 62 ? 32 : 40);
  const int alloc = (need + 7) / 8 * 8;
  const int w = 512 / alloc;
  return w > 8 ? 8 : (w < 2 ? 2 : w);
}

// ODD: the planes are 2*NSLOT-1 dwords long (else 2*NSLOT): the plane stride is a compile-time
// constant here, so every plane access is one instruction with an immediate offset.
template <int MAXP, int NSLOT, bool AGENT, bool ODD>
__global__ __launch_bounds__(64 * WAVES_PER_BLOCK, step_waves(MAXP, NSLOT)) void step_kernel(StepArgs A) {
  constexpr int FD = 2 * NSLOT - (ODD ? 1 : 0);
  constexpr int ROW_DW = (Planes<MAXP>::COUNT * FD + 3) / 4 * 4;
  __shared__ int32_t army_shadow[WAVES_PER_BLOCK][NSLOT * 64];  // per wave: the action phase's army copy
  using B = Turn<MAXP, NSLOT>;
  __shared__ uint32_t act_scratch[WAVES_PER_BLOCK][B::ACT_SCRATCH_DW];
  const int wave = (int)(threadIdx.x >> 6), lane = lane_id();
  const int env = uni((int)blockIdx.x * WAVES_PER_BLOCK + wave);
  if (env >= A.num_envs) return;
  bool force_redeal = false;
  if constexpr (!AGENT) {
    // GVEC_ACT_SKIP_ENV on player 0's action: this env sits the call out (nothing is read or
    // written; the host keeps the legal-mask buffer current for such calls)
    const uint32_t f0 = (uint32_t)uni((int)reinterpret_cast<const uint32_t*>(A.actions)[((size_t)env * A.pstride) * 2 + 1]);
    if (f0 & GVEC_ACT_SKIP_ENV) {
      if (A.err && lane == 0) A.err[env] = 0;
      return;
    }
    force_redeal = (f0 & GVEC_ACT_RESET_ENV) != 0u;  // the caller ends this episode (truncation): re-deal now
  }
  B b;
  b.larmy = army_shadow[wave];
  b.lscr = act_scratch[wave];
  const ArmyRef army_env = army_ref<NSLOT>(A.army16, A.army32, env);
  load_turn<true>(b, A.hdr + (size_t)env * HDR_DW, A.rows + (size_t)env * ROW_DW, army_env, FD, A.zeros);
  b.small = !(b.hflags & HF_WIDE);  // one turn from armies <= 65535: every sum of the turn stays below 2^23
  const bool emit = (A.flags & KF_EMIT) != 0u;
  uint32_t m[B::NR][4];
  uint32_t err = 0u;
  bool types_dirty = false, changed = true;
  const bool can_redeal = (A.flags & KF_AUTORESET) && A.pool_size > 0;
  if (AGENT && A.actions_out && (b.hflags & HF_DONE)) {
    // a re-dealt or frozen env plays no move in this launch: its recorded actions are "none", not whatever an
    // earlier recorded step left in the buffer (gvec_experience_records builds acted bits from these words)
    if (lane < A.pstride) reinterpret_cast<uint2*>(A.actions_out)[(size_t)env * A.pstride + lane] = make_uint2(0u, 0u);
  }
  if (can_redeal && ((b.hflags & HF_DONE) || force_redeal)) {
    redeal<MAXP, NSLOT>(b, A, env, FD, ROW_DW);  // the pool board brings its own gt1 plane
    types_dirty = true;
  } else if (b.hflags & HF_DONE) {
    err = GVEC_ERR_GAME_OVER;  // turn_processor.go:95-113: the engine stays frozen
    changed = false;
  } else {
    typename B::ActVec av;
    if constexpr (AGENT) {
      // the agent's input: the legal-move planes of the resident state, rebuilt from the stored gt1 plane
      // (7 vector instructions; re-reading the 832-byte masks the previous launch wrote would cost more)
      b.template legal_planes<false>(m);
      uint32_t mine = (GVEC_PROFILE_SKIP & 1) ? 0u : agent_sample<MAXP, NSLOT>(b, m, env_key_of(A.seed_base, (uint32_t)env), A);
      if (GVEC_PROFILE_DUP & 64) { b.opaque(); b.template legal_planes<false>(m); }
      if (GVEC_PROFILE_DUP & 1) {
        b.opaque();
#pragma unroll
        for (int d = 0; d < 4; ++d) asm volatile("" : "+v"(m[0][d]));
        const uint32_t again = agent_sample<MAXP, NSLOT>(b, m, env_key_of(A.seed_base, (uint32_t)env), A);
        asm volatile("" : : "v"(again));
      }
      av = agent_actvec<MAXP, NSLOT>(b, mine, A.invalid_permille > 0);
      if (A.actions_out) {
        uint32_t alo, ahi;
        agent_words<MAXP, NSLOT>(b, mine, alo, ahi);
        if (lane < A.pstride) reinterpret_cast<uint2*>(A.actions_out)[(size_t)env * A.pstride + lane] = make_uint2(alo, ahi);
      }
    } else {
      uint32_t alo = 0u, ahi = 0u;
      if (lane < A.pstride) {
        const uint2 w = reinterpret_cast<const uint2*>(A.actions)[(size_t)env * A.pstride + lane];
        alo = w.x;
        ahi = w.y;
      }
      av = b.prevalidate(alo, ahi);
    }
    bool aborted;
    err = b.turn_step(av, A, aborted);
    if (!(GVEC_PROFILE_SKIP & 32)) b.refresh_gt1();
    if (GVEC_PROFILE_DUP & 32) { b.opaque_v(); b.refresh_gt1(); }
    b.hdr_set(H_CNT_STEPS, b.hdr_get(H_CNT_STEPS) + 1u);
    if (aborted) b.hdr_set(H_CNT_ABORT, b.hdr_get(H_CNT_ABORT) + 1u);
    if (b.hflags & HF_DONE) b.hdr_set(H_CNT_DONE, b.hdr_get(H_CNT_DONE) + 1u);
  }
  b.store_army_staged(army_env);  // picks the narrow / wide form: before the header, which records it
  b.settle_lists();               // ... and so is whether the list planes are stored
  b.store_hdr(A.hdr + (size_t)env * HDR_DW, err);
  if (types_dirty) b.store_planes(A.rows + (size_t)env * ROW_DW, FD, ROW_DW, true);
  else b.store_planes_staged(A.rows + (size_t)env * ROW_DW, FD);
  if (A.err && lane == 0) A.err[env] = (int32_t)err;
  if (!(GVEC_PROFILE_SKIP & 64) && emit && (changed || !(A.flags & KF_LMVALID))) {
    b.template legal_planes<false>(m);
    b.store_masks_staged(m, A.legal + (size_t)env * A.pstride * A.mask_dw, FD, A.pstride);
  }
}

// Waves per SIMD to ask of the register allocator for the fused rollout kernel.  With no HBM traffic
// inside the turn loop the kernel is latency / issue bound, and occupancy pays even at the price of
// a few scratch spills; a spill inside the turn loop of the big variants costs far more than a wave.
constexpr int rollout_waves(int maxp, int nslot) {
  const int ppr = (nslot <= 7) ? 4 : 2;
  const int nr = (maxp + ppr - 1) / ppr;
  const int need = 3 * nr + 13 + nr + nslot + 4 * nr + (nslot >= 16 ? 100 : nslot >= 10 ? 68 : 44);
  const int alloc = (need + 7) / 8 * 8;
  const int w = 512 / alloc;
  return w > 8 ? 8 : (w < 2 ? 2 : w);
}

// `turns` engine turns per launch with the board kept in registers / LDS (fused rollouts; always
// with the on-device agent)
template <int MAXP, int NSLOT>
__global__ __launch_bounds__(64 * WAVES_PER_BLOCK, rollout_waves(MAXP, NSLOT)) void rollout_kernel(StepArgs A) {
  __shared__ int32_t army_shadow[WAVES_PER_BLOCK][NSLOT * 64];
  using B = Turn<MAXP, NSLOT>;
  __shared__ uint32_t act_scratch[WAVES_PER_BLOCK][B::ACT_SCRATCH_DW];
  const int wave = (int)(threadIdx.x >> 6), lane = lane_id();
  const int env = uni((int)blockIdx.x * WAVES_PER_BLOCK + wave);
  if (env >= A.num_envs) return;
  B b;
  b.larmy = army_shadow[wave];
  b.lscr = act_scratch[wave];
  const ArmyRef army_env = army_ref<NSLOT>(A.army16, A.army32, env);
  load_turn(b, A.hdr + (size_t)env * HDR_DW, A.rows + (size_t)env * A.row_dw, army_env, A.fd, A.zeros);
  uint32_t m[B::NR][4];
  uint32_t err = 0u, n_steps = 0u, n_abort = 0u, n_done = 0u;
  const uint32_t ek = env_key_of(A.seed_base, (uint32_t)env);
  const bool can_redeal = (A.flags & KF_AUTORESET) && A.pool_size > 0;
  int k = 0;
  // The hot inner loop plays turns while the game is live; the rare events (game over: re-deal from
  // the pool, or freeze) sit in the outer loop so they do not shape the inner loop's registers.
  for (;;) {
    b.template legal_planes<false>(m);  // the planes of the CURRENT state: the agent's input, the output at the end
    while (k < A.turns && !(b.hflags & HF_DONE)) {
      const uint32_t mine = agent_sample<MAXP, NSLOT>(b, m, ek, A);
      bool aborted;
      err = b.turn_step(agent_actvec<MAXP, NSLOT>(b, mine, A.invalid_permille > 0), A, aborted);
      n_steps += 1u;
      n_abort += aborted ? 1u : 0u;
      n_done += (b.hflags & HF_DONE) ? 1u : 0u;
      ++k;
      b.refresh_gt1();
      b.template legal_planes<false>(m);
    }
    if (k >= A.turns) break;
    if (!can_redeal) {
      err = GVEC_ERR_GAME_OVER;  // frozen for the rest of the launch
      break;
    }
    redeal<MAXP, NSLOT>(b, A, env, A.fd, A.row_dw);  // this turn slot is spent re-dealing (vector-env auto-reset)
    err = 0u;
    ++k;
  }
  b.hdr_set(H_CNT_STEPS, b.hdr_get(H_CNT_STEPS) + n_steps);
  b.hdr_set(H_CNT_ABORT, b.hdr_get(H_CNT_ABORT) + n_abort);
  b.hdr_set(H_CNT_DONE, b.hdr_get(H_CNT_DONE) + n_done);
  b.store_army(army_env);
  b.settle_lists();
  b.store_hdr(A.hdr + (size_t)env * HDR_DW, err);
  b.store_planes(A.rows + (size_t)env * A.row_dw, A.fd, A.row_dw, true);
  if (A.err && lane == 0) A.err[env] = (int32_t)err;
  b.store_masks_staged(m, A.legal + (size_t)env * A.pstride * A.mask_dw, A.fd, A.pstride);
}

// legal masks / agent actions of the resident state (no turn is played)
// MODE 0: Engine.GetLegalActionMask   1: random-agent actions   2: Serializer.GenerateActionMask
template <int MAXP, int NSLOT, int MODE>
__global__ __launch_bounds__(64 * WAVES_PER_BLOCK) void query_kernel(StepArgs A) {
  using B = Turn<MAXP, NSLOT>;
  const int wave = (int)(threadIdx.x >> 6), lane = lane_id();
  const int env = uni((int)blockIdx.x * WAVES_PER_BLOCK + wave);
  if (env >= A.num_envs) return;
  B b;
  b.larmy = nullptr;
  load_turn(b, A.hdr + (size_t)env * HDR_DW, A.rows + (size_t)env * A.row_dw, army_cref<NSLOT>(A.army16, A.army32, env), A.fd, A.zeros);
  uint32_t m[B::NR][4];
  if constexpr (MODE == 2) b.template legal_planes<true>(m);
  else b.template legal_planes<false>(m);
  if constexpr (MODE == 1) {
    uint32_t alo = 0u, ahi = 0u;
    if (!(b.hflags & HF_DONE)) agent_words<MAXP, NSLOT>(b, agent_sample<MAXP, NSLOT>(b, m, env_key_of(A.seed_base, (uint32_t)env), A), alo, ahi);
    if (lane < A.pstride) reinterpret_cast<uint2*>(A.actions_out)[(size_t)env * A.pstride + lane] = make_uint2(alo, ahi);
  } else {
    b.store_masks(m, A.legal + (size_t)env * A.pstride * A.mask_dw, A.fd, A.pstride);
  }
}

// EngineInitializer.performInitialSetup (engine_initializer.go:218-225) for the envs an import marked
// (HF_SETUP): full stats pass, full fog pass, game-over check on the freshly imported board.
template <int MAXP, int NSLOT>
__global__ __launch_bounds__(64 * WAVES_PER_BLOCK) void setup_kernel(ImportArgs A) {
  using B = Turn<MAXP, NSLOT>;
  const int wave = (int)(threadIdx.x >> 6);
  const int i = uni((int)blockIdx.x * WAVES_PER_BLOCK + wave);
  if (i >= A.n) return;
  const int env = A.env_ids ? uni(A.env_ids[i]) : A.dst_begin + i;
  if (env < 0 || env >= A.dst_envs) return;  // reported by the import kernel
  B b;
  b.larmy = nullptr;
  load_turn(b, A.hdr + (size_t)env * HDR_DW, A.rows + (size_t)env * A.row_dw, army_cref<NSLOT>(A.army16, A.army32, env), A.fd, A.zeros);
  if (!(b.hflags & HF_SETUP)) return;  // this env's input was rejected: left as it was
  b.hflags &= ~HF_SETUP;
  b.initial_setup();
  b.settle_lists();
  b.store_hdr(A.hdr + (size_t)env * HDR_DW, 0u);
  b.store_planes(A.rows + (size_t)env * A.row_dw, A.fd, A.row_dw, false);
}

// =========================================================================================
// internal/experience: snapshot (GameState.Clone before the step), rewards, observation tensor
// =========================================================================================
// Snapshot of env e (snap_dw dwords): what the reward AND the experience record need of the state before the
// step - prev own planes [MAXP][fd] | prev vis planes [MAXP][fd] | Serializer.GenerateActionMask(prev) as four
// direction planes per player [MAXP][4][fd] | prev armies as u16, tile t at halfword t, saturated to [0, 65535]
// (StateToTensor clamps army / 1000 at 1, serializer.go:82-85: saturation is exact for it) [NSLOT*32] |
// tail: territory [MAXP], armies [MAXP], turn, W | H << 8.
template <int MAXP, int NSLOT>
struct SnapLayout {
  int fd;
  __host__ __device__ int own() const { return 0; }
  __host__ __device__ int vis() const { return MAXP * fd; }
  __host__ __device__ int mask() const { return 2 * MAXP * fd; }
  __host__ __device__ int army() const { return 6 * MAXP * fd; }
  __host__ __device__ int tail() const { return 6 * MAXP * fd + NSLOT * 32; }
  __host__ __device__ int total() const { return (tail() + 2 * MAXP + 2 + 3) / 4 * 4; }
};

// Tile.Army as the u16 a tensor consumer needs: lane l of slot s is tile 64s + l
template <int NSLOT>
__device__ __forceinline__ void store_army_sat16(const int32_t (&army)[NSLOT], uint32_t* dst) {
  uint16_t* h = reinterpret_cast<uint16_t*>(dst);
#pragma unroll
  for (int s = 0; s < NSLOT; ++s) {
    const int32_t a = army[s];
    h[64 * s + lane_id()] = (uint16_t)(a < 0 ? 0 : (a > 65535 ? 65535 : a));
  }
}

template <int MAXP, int NSLOT>
__global__ __launch_bounds__(64 * WAVES_PER_BLOCK) void snapshot_kernel(ExperienceArgs A) {
  using B = Board<MAXP, NSLOT>;
  const int wave = (int)(threadIdx.x >> 6), lane = lane_id();
  const int i = uni((int)blockIdx.x * WAVES_PER_BLOCK + wave);
  if (i >= A.num_envs) return;
  const int env = A.env_begin + i;
  B b;
  load_board(b, A.hdr + (size_t)env * HDR_DW, A.rows + (size_t)env * A.row_dw, army_cref<NSLOT>(A.army16, A.army32, env), A.fd);
  const SnapLayout<MAXP, NSLOT> L{A.fd};
  uint32_t* sn = A.snap + (size_t)env * A.snap_dw;
  uint32_t tail = 0u;  // lane p: territory, lane MAXP+p: armies, lane 2*MAXP: turn, +1: W|H<<8
#pragma unroll
  for (int p = 0; p < MAXP; ++p) {
    if (lane < A.fd) {
      sn[L.own() + p * A.fd + lane] = b.own[p];
      sn[L.vis() + p * A.fd + lane] = b.vis[p];
      // Serializer.GenerateActionMask (serializer.go:112-176): board owner, army >= 2, no Alive check;
      // d = 0 up, 1 DOWN, 2 LEFT, 3 right
      const uint32_t src = b.own[p] & b.gt1;
      uint32_t* m = sn + L.mask() + p * 4 * A.fd + lane;
      m[0 * A.fd] = src & b.ok[0];
      m[1 * A.fd] = src & b.ok[2];
      m[2 * A.fd] = src & b.ok[3];
      m[3 * A.fd] = src & b.ok[1];
    }
    const int32_t terr = b.count(b.own[p]), arm = b.army_sum(b.own[p]);
    tail = (lane == p) ? (uint32_t)terr : tail;
    tail = (lane == MAXP + p) ? (uint32_t)arm : tail;
  }
  store_army_sat16<NSLOT>(b.army, sn + L.army());
  tail = (lane == 2 * MAXP) ? (uint32_t)b.turn : tail;
  tail = (lane == 2 * MAXP + 1) ? ((uint32_t)b.W | ((uint32_t)b.H << 8)) : tail;
  if (lane < 2 * MAXP + 2) sn[L.tail() + lane] = tail;
}

// CalculateRewardWithConfig (internal/experience/rewards.go:45-85) with DefaultRewardConfig (:23-37);
// prev = the snapshot, cur = the resident state.  float32 arithmetic in the reference's order,
// compiled with -ffp-contract=off (Go on amd64 does not fuse multiply-add).  Lane p receives player p's reward.
template <int MAXP, int NSLOT>
__device__ __forceinline__ float compute_rewards(const Board<MAXP, NSLOT>& b, const uint32_t* sn, int fd, bool& over, bool& comparable) {
  const int lane = lane_id();
  const SnapLayout<MAXP, NSLOT> L{fd};
  const uint32_t tail = (lane < 2 * MAXP + 2) ? sn[L.tail() + lane] : 0u;
  const int prev_turn = (int)rdlane(tail, 2 * MAXP);
  const uint32_t prev_dims = rdlane(tail, 2 * MAXP + 1);
  // a board re-dealt by auto-reset (or not stepped) has no meaningful predecessor: reward 0
  comparable = prev_dims == ((uint32_t)b.W | ((uint32_t)b.H << 8)) && b.turn > prev_turn;
  const int na = __builtin_popcount(b.alive);
  over = na <= 1;                                                // GameState.IsGameOver (state.go:73-82)
  const int winner = (na == 1) ? (31 - __builtin_clz(b.alive)) : -1;  // GameState.GetWinner (state.go:85-100)
  uint32_t prev_own[MAXP], prev_any = 0u;
  int32_t cur_arm[MAXP], total = 0;
#pragma unroll
  for (int p = 0; p < MAXP; ++p) {
    prev_own[p] = (lane < fd) ? sn[L.own() + p * fd + lane] : 0u;
    prev_any |= prev_own[p];
    cur_arm[p] = b.army_sum(b.own[p]);
    total += cur_arm[p];
  }
  float rv = 0.0f;
#pragma unroll
  for (int p = 0; p < MAXP; ++p) {
    const int d_terr = b.count(b.own[p]) - (int)rdlane(tail, p);             // :59-62
    const int d_arm = cur_arm[p] - (int)rdlane(tail, MAXP + p);               // :65-68
    const int c_gain = b.count(b.city & b.own[p] & ~prev_own[p]);             // countCityChanges :110-129
    const int c_lost = b.count(b.city & prev_own[p] & ~b.own[p]);
    const int g_gain = b.count(b.gen & b.own[p] & ~prev_own[p] & prev_any);   // countGeneralChanges :132-151
    const int g_lost = b.count(b.gen & prev_own[p] & ~b.own[p]);
    float r = 0.0f;
    r += (float)d_terr * 0.01f;
    r += (float)d_arm * 0.001f;
    r += (float)c_gain * 0.1f;
    r += (float)c_lost * -0.1f;
    r += (float)g_gain * 0.5f;
    r += (float)g_lost * -0.5f;
    const int pa = cur_arm[p], ea = total - cur_arm[p];                       // calculateArmyAdvantage :153-175
    const float adv = (total == 0) ? 0.0f : ((float)(pa - ea) / (float)total);
    r += adv * 0.05f;
    if (over && winner == p) r = 1.0f;                                        // :49-56
    else if (over && winner != -1) r = -1.0f;
    r = (comparable && p < b.P) ? r : 0.0f;
    rv = (lane == p) ? r : rv;
  }
  return rv;
}

template <int MAXP, int NSLOT>
__global__ __launch_bounds__(64 * WAVES_PER_BLOCK) void rewards_kernel(ExperienceArgs A) {
  using B = Board<MAXP, NSLOT>;
  const int wave = (int)(threadIdx.x >> 6), lane = lane_id();
  const int i = uni((int)blockIdx.x * WAVES_PER_BLOCK + wave);
  if (i >= A.num_envs) return;
  const int env = A.env_begin + i;
  B b;
  load_board(b, A.hdr + (size_t)env * HDR_DW, A.rows + (size_t)env * A.row_dw, army_cref<NSLOT>(A.army16, A.army32, env), A.fd);
  bool over, comparable;
  const float rv = compute_rewards<MAXP, NSLOT>(b, A.snap + (size_t)env * A.snap_dw, A.fd, over, comparable);
  if (lane < A.pstride) A.rewards[(size_t)i * A.pstride + lane] = rv;
  if (A.done && lane == 0) A.done[i] = (uint8_t)(over ? 1 : 0);
}

// One EXPERIENCE RECORD per env transition: everything SimpleCollector.OnStateTransition (internal/experience/
// collector.go:30-98) puts into the experiencepb.Experience of every player that acted, in compact form -
// bit-planes and u16 armies instead of 2 x P x [9][H][W] float tensors (3.7 KB instead of 115 KB at 20x20 4P): what a
// rank ships over xGMI to the process that feeds StreamAggregator, which expands it (experience.decode_records).
//   dword 0 currState.Turn | 1 W | H<<8 | P<<16 | flags<<24 (1 done = currState.IsGameOver, 2 fog of war, 4 valid: the env
//   was not re-dealt) | 2 acted bits (players that submitted an action, collector.go:33-37) | 3 env id
//   4.. action index per player (Serializer.ActionToIndex, serializer.go:179-198; -1: none) | rewards f32 per player
//   planes [fd]: prev own[MAXP], prev vis[MAXP], next own[MAXP], next vis[MAXP], general, city, mountain
//   GenerateActionMask(prev) [MAXP][4][fd] | prev armies u16 [NSLOT*64] | next armies u16 [NSLOT*64]
template <int MAXP, int NSLOT>
struct RecordLayout {
  int fd;
  __host__ __device__ int action() const { return 4; }
  __host__ __device__ int reward() const { return 4 + MAXP; }
  __host__ __device__ int planes() const { return 4 + 2 * MAXP; }
  __host__ __device__ int mask() const { return planes() + (4 * MAXP + 3) * fd; }
  __host__ __device__ int army_prev() const { return mask() + 4 * MAXP * fd; }
  __host__ __device__ int army_next() const { return army_prev() + NSLOT * 32; }
  __host__ __device__ int total() const { return (army_next() + NSLOT * 32 + 3) / 4 * 4; }
};

template <int MAXP, int NSLOT>
__global__ __launch_bounds__(64 * WAVES_PER_BLOCK) void experience_record_kernel(ExperienceArgs A) {
  using B = Board<MAXP, NSLOT>;
  const int wave = (int)(threadIdx.x >> 6), lane = lane_id();
  const int i = uni((int)blockIdx.x * WAVES_PER_BLOCK + wave);
  if (i >= A.num_envs) return;
  const int env = A.env_begin + i;
  B b;
  load_board(b, A.hdr + (size_t)env * HDR_DW, A.rows + (size_t)env * A.row_dw, army_cref<NSLOT>(A.army16, A.army32, env), A.fd);
  const uint32_t* sn = A.snap + (size_t)env * A.snap_dw;
  const SnapLayout<MAXP, NSLOT> S{A.fd};
  const RecordLayout<MAXP, NSLOT> R{A.fd};
  uint32_t* rec = A.records + (size_t)i * A.record_dw;
  bool over, comparable;
  const float rv = compute_rewards<MAXP, NSLOT>(b, sn, A.fd, over, comparable);
  // lane p: player p's action -> Serializer.ActionToIndex(action, prevState.Board.W)
  uint32_t alo = 0u, ahi = 0u;
  if (lane < A.pstride) {
    const uint2 w = reinterpret_cast<const uint2*>(A.actions)[(size_t)env * A.pstride + lane];
    alo = w.x;
    ahi = w.y;
  }
  const int prev_w = (int)(rdlane((lane < 2 * MAXP + 2) ? sn[S.tail() + lane] : 0u, 2 * MAXP + 1) & 0xFFu);
  const int fx = (int)(int8_t)(alo & 0xFFu), fy = (int)(int8_t)((alo >> 8) & 0xFFu);
  const int dx = (int)(int8_t)((alo >> 16) & 0xFFu) - fx, dy = (int)(int8_t)(alo >> 24) - fy;
  int dir = 0;                                   // :183-195: up 0 (and anything that is not a unit step), down 1, left 2, right 3
  dir = (dx == 0 && dy == 1) ? 1 : dir;
  dir = (dx == -1 && dy == 0) ? 2 : dir;
  dir = (dx == 1 && dy == 0) ? 3 : dir;
  const bool acted = lane < b.P && (ahi & GVEC_ACT_VALID) != 0u;
  const int aidx = acted ? ((fy * prev_w + fx) * 4 + dir) : -1;
  const uint32_t acted_bits = (uint32_t)__builtin_amdgcn_ballot_w64(acted);
  if (lane < MAXP) {
    rec[R.action() + lane] = (uint32_t)aidx;
    rec[R.reward() + lane] = __float_as_uint(rv);
  }
  if (lane == 0) {
    rec[0] = (uint32_t)b.turn;
    rec[1] = (uint32_t)b.W | ((uint32_t)b.H << 8) | ((uint32_t)b.P << 16) |
             (((over ? 1u : 0u) | ((b.hflags & HF_FOG) ? 2u : 0u) | (comparable ? 4u : 0u)) << 24);
    rec[2] = acted_bits;
    rec[3] = (uint32_t)(A.env_id_base + env);
  }
  if (lane < A.fd) {
    uint32_t* pl = rec + R.planes() + lane;
#pragma unroll
    for (int p = 0; p < MAXP; ++p) {
      pl[p * A.fd] = sn[S.own() + p * A.fd + lane];
      pl[(MAXP + p) * A.fd] = sn[S.vis() + p * A.fd + lane];
      pl[(2 * MAXP + p) * A.fd] = b.own[p];
      pl[(3 * MAXP + p) * A.fd] = b.vis[p];
    }
    pl[(4 * MAXP + 0) * A.fd] = b.gen;
    pl[(4 * MAXP + 1) * A.fd] = b.city;
    pl[(4 * MAXP + 2) * A.fd] = b.mtn;
  }
  for (int k = lane; k < 4 * MAXP * A.fd; k += 64) rec[R.mask() + k] = sn[S.mask() + k];
  for (int k = lane; k < NSLOT * 32; k += 64) rec[R.army_prev() + k] = sn[S.army() + k];
  store_army_sat16<NSLOT>(b.army, rec + R.army_next());
  for (int k = R.army_next() + NSLOT * 32 + lane; k < A.record_dw; k += 64) rec[k] = 0u;
}

// The consumer side of the exchange step: expands compact experience records - on whatever GPU they were
// gathered to - into what SimpleCollector.OnStateTransition (collector.go:41-75) puts into every acting player's
// experiencepb.Experience: StateToTensor(prevState, p), StateToTensor(currState, p) ([9][H][W] float32, serializer.go:37-109),
// GenerateActionMask(prevState, p) ([]bool, index t*4 + d, d = 0 up, 1 down, 2 left, 3 right, :112-176), and the scalar
// fields.  One wavefront per record; needs no engine handle (the record carries its own W, H, P, flags; the layout
// constants arrive as arguments).  A u16 army saturated at 65,535 is exact here: the tensor clamps army / 1000 at 1.
struct ExpandArgs {
  const uint32_t* records;  // [n][record_dw]
  float* state;             // [n][mp][9*stride]
  float* next_state;        // [n][mp][9*stride]
  uint8_t* mask;            // [n][mp][4*stride] 0/1 bytes
  int32_t* meta;            // [n][mp][8]: present (valid record & the player acted), env id, player, turn, action, reward bits, done, W | H << 8
  int32_t n, record_dw, mp, fd, ns, stride;
};

// own / vis / types / army: this wave's LDS copy of the record; any: OR of the P ownership planes
// One StateToTensor of an expanded record.  Its nine planes are N = W*H floats each, back to back: for most boards no plane
// starts on a 256-byte boundary (15x15: 900 bytes; 20x20: 1,600), and stores of "tiles 64j .. 64j+63" that begin anywhere
// in a line leave at a third of the rate of aligned ones (32,768 records: 15x15 and 25x25 expanded at 1.9 TB/s, 16x16 and
// 32x32 at 5.9-6.3).  So every store covers an ALIGNED window of 64 floats: lane l of window k holds tile 64k - sh + l of its
// plane (sh: the plane's dword offset inside its 256-byte line); the bits come from the record in LDS, where any tile is
// as near as any other.
__device__ __forceinline__ void expand_tensor(float* out, const uint32_t* own_p, const uint32_t* any, const uint32_t* vis_p, const uint32_t* types,
                                              const uint16_t* army, int fd, int N, int stride, bool fog) {
  const int lane = lane_id();
  const uint32_t a0 = (uint32_t)(reinterpret_cast<uintptr_t>(out) >> 2);
  if (((a0 | (uint32_t)N) & 3u) == 0u) {
    // planes of a multiple of four floats on 16-byte boundaries (10x10, 16x16, 20x20, 32x32 ...): FOUR neighbouring tiles per
    // lane - a nibble of each of the record's planes, four u16 armies in one LDS read, nine 1-KB stores per 256 tiles, which
    // (unlike 256-byte runs of dwords) leave at full rate wherever they start
    auto nib = [&](const uint32_t* plane, int q) { return (plane[q >> 3] >> ((q & 7) << 2)) & 15u; };
    for (int q = lane; q < (N >> 2); q += 64) {
      const uint32_t n_any = nib(any, q), n_mine = nib(own_p, q), n_seen = nib(vis_p, q);
      const uint32_t n_spec = nib(types, q) | nib(types + fd, q), n_mtn = nib(types + 2 * fd, q);
      const uint32_t n_vis = fog ? n_seen : 15u;                   // :50
      const uint32_t n_open = n_vis & ~n_mtn;                      // mountains short-circuit (:68-71)
      const uint32_t* ap = reinterpret_cast<const uint32_t*>(army) + 2 * q;    // the record's armies are dword-aligned in LDS
      const uint2 aw = make_uint2(ap[0], ap[1]);
      auto arm = [](uint32_t a) {
        float norm = (float)(int)a / 1000.0f;                      // :82-85
        norm = norm > 1.0f ? 1.0f : norm;
        return (a > 0u) ? norm : 0.0f;
      };
      const float r0 = arm(aw.x & 0xFFFFu), r1 = arm(aw.x >> 16), r2 = arm(aw.y & 0xFFFFu), r3 = arm(aw.y >> 16);
      const size_t n = (size_t)N;
      auto put = [&](int c, uint32_t m, float v0, float v1, float v2, float v3) {
        float4 o;
        o.x = (m & 1u) ? v0 : 0.0f;
        o.y = (m & 2u) ? v1 : 0.0f;
        o.z = (m & 4u) ? v2 : 0.0f;
        o.w = (m & 8u) ? v3 : 0.0f;
        st_stream<GVEC_NT_MASK>(reinterpret_cast<u32x4*>(out + c * n) + q, *reinterpret_cast<const u32x4*>(&o));
      };
      const uint32_t m_mine = n_open & n_mine, m_other = n_open & ~n_mine & n_any;
      put(0, m_mine, r0, r1, r2, r3);
      put(1, m_other, r0, r1, r2, r3);
      put(2, m_mine, 1.0f, 1.0f, 1.0f, 1.0f);
      put(3, m_other, 1.0f, 1.0f, 1.0f, 1.0f);
      put(4, n_open & ~n_any, 1.0f, 1.0f, 1.0f, 1.0f);
      put(5, n_open & n_spec, 1.0f, 1.0f, 1.0f, 1.0f);
      put(6, n_vis & n_mtn, 1.0f, 1.0f, 1.0f, 1.0f);
      put(7, n_vis, 1.0f, 1.0f, 1.0f, 1.0f);
      put(8, ~n_vis & 15u, 1.0f, 1.0f, 1.0f, 1.0f);
    }
  } else
#pragma unroll
  for (int c = 0; c < 9; ++c) {
    float* plane = out + (size_t)c * (size_t)N;
    const int sh = (int)((a0 + (uint32_t)c * (uint32_t)N) & 63u);
    for (int t = lane - sh; t < N; t += 64) {
      if (t < 0) continue;
      const int dwi = t >> 5;
      const uint32_t bit = 1u << (t & 31);
      const bool seen = (vis_p[dwi] & bit) != 0u, mount = (types[2 * fd + dwi] & bit) != 0u;
      const bool visible = !fog || seen;      // :50
      const bool open = visible && !mount;    // mountains short-circuit (:68-71)
      float v;
      if (c <= 3) {
        const bool owned = (any[dwi] & bit) != 0u, mine = (own_p[dwi] & bit) != 0u;
        const bool who = (c & 1) ? (!mine && owned) : mine;      // 0, 2: the player's own; 1, 3: somebody else's
        if (c < 2) {
          const int a = (int)army[t];
          float norm = (float)a / 1000.0f;    // :82-85
          norm = norm > 1.0f ? 1.0f : norm;
          v = (open && who && a > 0) ? norm : 0.0f;
        } else {
          v = (open && who) ? 1.0f : 0.0f;
        }
      } else if (c == 4) {
        v = (open && !(any[dwi] & bit)) ? 1.0f : 0.0f;
      } else if (c == 5) {
        v = (open && ((types[dwi] | types[fd + dwi]) & bit)) ? 1.0f : 0.0f;
      } else if (c == 6) {
        v = (visible && mount) ? 1.0f : 0.0f;
      } else {
        v = (visible == (c == 7)) ? 1.0f : 0.0f;
      }
      st_stream<GVEC_NT_MASK>(plane + t, v);
    }
  }
  for (int i = 9 * N + lane; i < 9 * stride; i += 64) out[i] = 0.0f;  // a smaller board in a padded batch: clear the rest of the slot
}

// dynamic LDS: per wave the record (record_dw dwords) + two fd-dword "anybody owns it" planes (prev, next)
__global__ __launch_bounds__(64 * WAVES_PER_BLOCK) void expand_records_kernel(ExpandArgs A) {
  extern __shared__ uint32_t expand_lds[];
  const int wave = (int)(threadIdx.x >> 6), lane = lane_id();
  const int i = uni((int)blockIdx.x * WAVES_PER_BLOCK + wave);
  if (i >= A.n) return;
  const int mp = A.mp, fd = A.fd;
  uint32_t* rec = expand_lds + (size_t)wave * (A.record_dw + 2 * fd);
  {
    const uint32_t* g = A.records + (size_t)i * A.record_dw;
    for (int k = lane; k < A.record_dw; k += 64) rec[k] = g[k];
  }
  wave_lds_fence();
  const uint32_t r1 = rec[1];
  int W = (int)(r1 & 0xFFu), H = (int)((r1 >> 8) & 0xFFu), P = (int)((r1 >> 16) & 0xFFu);
  const uint32_t flags = r1 >> 24;
  const bool ok = W >= 1 && H >= 1 && W * H <= A.stride && P >= 1 && P <= mp;   // a malformed record expands to nothing
  if (!ok) W = H = P = 0;
  const int N = W * H;
  const uint32_t acted = ok && (flags & 4u) ? rec[2] : 0u;   // a void record (re-dealt env) yields no experience
  const int off_planes = 4 + 2 * mp, off_mask = off_planes + (4 * mp + 3) * fd, off_prev = off_mask + 4 * mp * fd, off_next = off_prev + A.ns * 32;
  const uint32_t* prev_own = rec + off_planes;
  const uint32_t* prev_vis = prev_own + mp * fd;
  const uint32_t* next_own = prev_vis + mp * fd;
  const uint32_t* next_vis = next_own + mp * fd;
  const uint32_t* types = next_vis + mp * fd;   // general, city, mountain
  const uint16_t* army_prev = reinterpret_cast<const uint16_t*>(rec + off_prev);
  const uint16_t* army_next = reinterpret_cast<const uint16_t*>(rec + off_next);
  uint32_t* any_prev = rec + A.record_dw;
  uint32_t* any_next = any_prev + fd;
  if (lane < fd) {
    uint32_t a = 0u, b = 0u;
    for (int q = 0; q < P; ++q) {
      a |= prev_own[q * fd + lane];
      b |= next_own[q * fd + lane];
    }
    any_prev[lane] = a;
    any_next[lane] = b;
  }
  wave_lds_fence();
  for (int p = 0; p < mp; ++p) {
    const size_t slot = (size_t)i * mp + p;
    const bool present = p < P && ((acted >> p) & 1u) != 0u;
    int32_t* meta = A.meta + slot * 8;
    if (lane < 8) {
      int32_t v = 0;
      v = (lane == 0) ? (present ? 1 : 0) : v;
      v = (lane == 1) ? (int32_t)rec[3] : v;
      v = (lane == 2) ? p : v;
      v = (lane == 3) ? (int32_t)rec[0] : v;
      v = (lane == 4) ? (int32_t)rec[4 + p] : v;
      v = (lane == 5) ? (int32_t)rec[4 + mp + p] : v;
      v = (lane == 6) ? (int32_t)(flags & 1u) : v;
      v = (lane == 7) ? (int32_t)((uint32_t)W | ((uint32_t)H << 8)) : v;
      meta[lane] = present ? v : ((lane == 2) ? p : 0);
    }
    float* st = A.state + slot * 9 * (size_t)A.stride;
    float* nx = A.next_state + slot * 9 * (size_t)A.stride;
    uint8_t* mk = A.mask + slot * 4 * (size_t)A.stride;
    if (!present) {  // wave-uniform
      for (int k = lane; k < 9 * A.stride; k += 64) st[k] = nx[k] = 0.0f;
      for (int k = lane; k < A.stride; k += 64) reinterpret_cast<uint32_t*>(mk)[k] = 0u;
      continue;
    }
    expand_tensor(st, prev_own + p * fd, any_prev, prev_vis + p * fd, types, army_prev, fd, N, A.stride, (flags & 2u) != 0u);
    expand_tensor(nx, next_own + p * fd, any_next, next_vis + p * fd, types, army_next, fd, N, A.stride, (flags & 2u) != 0u);
    // GenerateActionMask as bytes, four per tile (t*4 + d): one dword store per tile
    const uint32_t* m = rec + off_mask + p * 4 * fd;   // [d][fd]
    const int msh = (int)((reinterpret_cast<uintptr_t>(mk) >> 2) & 63u);       // aligned windows here too
    for (int t = lane - msh; t < A.stride; t += 64) {
      if (t < 0) continue;
      uint32_t v = 0u;
      if (t < N) {
        const int dwi = t >> 5, sh = t & 31;
        v = ((m[dwi] >> sh) & 1u) | (((m[fd + dwi] >> sh) & 1u) << 8) | (((m[2 * fd + dwi] >> sh) & 1u) << 16) | (((m[3 * fd + dwi] >> sh) & 1u) << 24);
      }
      st_stream<GVEC_NT_MASK>(reinterpret_cast<uint32_t*>(mk) + t, v);
    }
  }
}

// Serializer.StateToTensor (internal/experience/serializer.go:37-109): [9][H][W] float32 from one
// player's perspective; the output is 9 coalesced channel planes per 64-tile slot.
template <int MAXP, int NSLOT>
__global__ __launch_bounds__(64 * WAVES_PER_BLOCK) void observe_kernel(ExperienceArgs A) {
  using B = Board<MAXP, NSLOT>;
  const int wave = (int)(threadIdx.x >> 6), lane = lane_id();
  const int env = uni((int)blockIdx.x * WAVES_PER_BLOCK + wave);
  if (env >= A.num_envs) return;
  B b;
  load_board(b, A.hdr + (size_t)env * HDR_DW, A.rows + (size_t)env * A.row_dw, army_cref<NSLOT>(A.army16, A.army32, env), A.fd);
  const bool fog_on = (b.hflags & HF_FOG) != 0u;
  uint32_t own_any = 0u;
#pragma unroll
  for (int p = 0; p < MAXP; ++p) own_any |= b.own[p];
  const uint32_t special = b.gen | b.city;
  const int p_lo = (A.player < 0) ? 0 : A.player, p_hi = (A.player < 0) ? A.pstride : A.player + 1;
  for (int pl = p_lo; pl < p_hi; ++pl) {
    float* out = A.obs + ((A.player < 0) ? ((size_t)env * A.pstride + pl) : (size_t)env) * 9 * (size_t)A.stride;
    uint32_t own_p = 0u, vis_p = 0u;
#pragma unroll
    for (int p = 0; p < MAXP; ++p) {
      own_p = (p == pl) ? b.own[p] : own_p;
      vis_p = (p == pl) ? b.vis[p] : vis_p;
    }
    const uint32_t a0 = (uint32_t)(reinterpret_cast<uintptr_t>(out) >> 2);
    if (((a0 | (uint32_t)b.N) & 63u) != 0u) {
      // planes that do not start on 256-byte boundaries (all boards but 16x16, 32x32 ...): aligned 64-float store windows, as
      // in gym_emit below.  65,536 envs, one player: 15x15 0.297 -> 0.157 ms, 25x25 0.790 -> 0.365, 10x10 0.110 -> 0.079,
      // 20x20 0.200 -> 0.170 (5.6 TB/s)
      // (every at() / army_at() is a ds_bpermute: evaluated by all lanes, never behind a lane-dependent `&&`)
      auto at = [&](uint32_t plane, int t) { return __builtin_amdgcn_ubfe(bperm((t >> 5) << 2, plane), (uint32_t)(t & 31), 1u) != 0u; };
      auto emit = [&](int c, auto&& value) {
        float* base = out + (size_t)c * (size_t)b.N;
        const int sh = (int)((a0 + (uint32_t)c * (uint32_t)b.N) & 63u);
#pragma unroll
        for (int k = 0; k <= NSLOT; ++k) {
          if (64 * k - sh < b.N) {                   // wave-uniform
            const int t = 64 * k - sh + lane;
            const bool ok = t >= 0 && t < b.N;
            const int tt = ok ? t : 0;
            const bool seen = at(vis_p, tt), mount = at(b.mtn, tt);
            const bool visible = !fog_on || seen;    // :50
            const float v = value(k, tt, sh, visible, visible && !mount, mount);   // open: mountains short-circuit (:68-71)
            if (ok) st_stream<GVEC_NT_MASK>(base + t, v);
          }
        }
      };
      auto arm_at = [&](int k, int sh) {             // tile 64k - sh + l: slot k (lanes l >= sh) or k - 1, sh lanes further on
        const int from = ((lane - sh) & 63) << 2;
        const int32_t cur = (int32_t)bperm(from, (uint32_t)b.army[k < NSLOT ? k : NSLOT - 1]);
        const int32_t prv = (int32_t)bperm(from, (uint32_t)b.army[k > 0 ? k - 1 : 0]);
        const int32_t a = (lane >= sh) ? cur : prv;
        float norm = (float)a / 1000.0f;             // :82-85
        norm = norm > 1.0f ? 1.0f : norm;
        return (a > 0) ? norm : 0.0f;
      };
      emit(0, [&](int k, int t, int sh, bool, bool open, bool) { const bool mine = at(own_p, t); const float arm = arm_at(k, sh); return (open && mine) ? arm : 0.0f; });
      emit(1, [&](int k, int t, int sh, bool, bool open, bool) {
        const bool mine = at(own_p, t), owned = at(own_any, t);
        const float arm = arm_at(k, sh);
        return (open && !mine && owned) ? arm : 0.0f;
      });
      emit(2, [&](int, int t, int, bool, bool open, bool) { const bool mine = at(own_p, t); return (open && mine) ? 1.0f : 0.0f; });
      emit(3, [&](int, int t, int, bool, bool open, bool) { const bool mine = at(own_p, t), owned = at(own_any, t); return (open && !mine && owned) ? 1.0f : 0.0f; });
      emit(4, [&](int, int t, int, bool, bool open, bool) { const bool owned = at(own_any, t); return (open && !owned) ? 1.0f : 0.0f; });
      emit(5, [&](int, int t, int, bool, bool open, bool) { const bool spec = at(special, t); return (open && spec) ? 1.0f : 0.0f; });
      emit(6, [&](int, int, int, bool visible, bool, bool mount) { return (visible && mount) ? 1.0f : 0.0f; });
      emit(7, [&](int, int, int, bool visible, bool, bool) { return visible ? 1.0f : 0.0f; });
      emit(8, [&](int, int, int, bool visible, bool, bool) { return visible ? 0.0f : 1.0f; });
    } else
#pragma unroll
    for (int s = 0; s < NSLOT; ++s) {
      const int t = 64 * s + lane;
      const bool mine = b.gather(own_p, s) != 0u, owned = b.gather(own_any, s) != 0u, seen = b.gather(vis_p, s) != 0u;
      const bool spec = b.gather(special, s) != 0u, mount = b.gather(b.mtn, s) != 0u;
      const bool visible = !fog_on || seen;  // :50
      const bool open = visible && !mount;   // mountains short-circuit (:68-71)
      float norm = (float)b.army[s] / 1000.0f;  // :82-85
      norm = norm > 1.0f ? 1.0f : norm;
      const float arm = (b.army[s] > 0) ? norm : 0.0f;
      if (t < b.N) {
        const size_t n = (size_t)b.N;
        // 14.4 KB per (env, player) that the kernel never reads back: streamed past the L2 like the turn's own stores
        st_stream<GVEC_NT_MASK>(out + 0 * n + t, (open && mine) ? arm : 0.0f);
        st_stream<GVEC_NT_MASK>(out + 1 * n + t, (open && !mine && owned) ? arm : 0.0f);
        st_stream<GVEC_NT_MASK>(out + 2 * n + t, (open && mine) ? 1.0f : 0.0f);
        st_stream<GVEC_NT_MASK>(out + 3 * n + t, (open && !mine && owned) ? 1.0f : 0.0f);
        st_stream<GVEC_NT_MASK>(out + 4 * n + t, (open && !owned) ? 1.0f : 0.0f);
        st_stream<GVEC_NT_MASK>(out + 5 * n + t, (open && spec) ? 1.0f : 0.0f);
        st_stream<GVEC_NT_MASK>(out + 6 * n + t, (visible && mount) ? 1.0f : 0.0f);
        st_stream<GVEC_NT_MASK>(out + 7 * n + t, visible ? 1.0f : 0.0f);
        st_stream<GVEC_NT_MASK>(out + 8 * n + t, visible ? 0.0f : 1.0f);
      }
    }
    // a smaller board in a padded batch: clear the rest of the slot
    for (int i = 9 * b.N + lane; i < 9 * A.stride; i += 64) out[i] = 0.0f;
  }
}

// =========================================================================================
// python/generals_gym/generals_env.py on the device: what GeneralsEnv builds from the GameState proto the
// server sends for its player token - observation (:291-342), valid-action mask (:344-387), reward
// (:499-561) - computed straight from the resident state with the proto's fog rules applied in the kernel
// (internal/grpc/gameserver/server.go:556-582: a tile that is neither visible nor "known in fog" shows type
// NORMAL / owner -1 / army 0; a fogged tile keeps its type, hides owner and army.  A hidden tile is a normal
// tile by definition (visibility_optimized.go:189-191), so the shown type is always the real one).
// =========================================================================================
// The observation and mask of one env, from replicated flat planes (row 0 is read) of either board layout: `seen` what
// the proto shows as visible, own_p / own_any the learner's and anybody's tiles, m0..m3 the four direction planes of
// _get_valid_actions_mask and `many` their OR (index 4: "a half move is valid iff a full move is").  ms: this wave's
// LDS stage of (NSLOT*64*5 + 15)/16*16 bytes.
template <int NSLOT, typename BT>
__device__ __forceinline__ void gym_emit(const BT& b, uint32_t seen, uint32_t own_p, uint32_t own_any, uint32_t m0, uint32_t m1, uint32_t m2,
                                         uint32_t m3, uint32_t many, float tc, float* obs, uint8_t* mask, uint8_t* ms, int stride) {
  const int lane = lane_id();
  const uint32_t a0 = (uint32_t)(reinterpret_cast<uintptr_t>(obs) >> 2);
  if (((a0 | (uint32_t)stride) & 3u) != 0u) {
    // Planes of W*H floats that do not even start on 16-byte boundaries (odd W*H - 15x15: 900 bytes at multiples of 900;
    // 25x25: 2,500): a store of "tiles 64s .. 64s+63" begins anywhere in a line, and such stores leave at a third of the rate
    // of aligned ones (65,536 envs: the nine planes of 25x25 took 0.53 ms, those of 32x32 - 1.6x the bytes - 0.21).  So every
    // store covers an ALIGNED window of 64 floats instead, lane l of window k holding tile 64k - sh + l of its plane (sh: the
    // plane's dword offset inside its 256-byte line), whose bits come with the same ds_bpermute a slot's would: 25x25 0.77 ->
    // 0.47 ms, 15x15 0.256 -> 0.232.  (Planes on 16-byte boundaries are better off with the stores below - 10x10 0.140 vs
    // 0.162 ms this way, 20x20 0.260 vs 0.283 - whose per-slot bits all nine planes share.)
    auto at = [&](uint32_t plane, int t) { return __builtin_amdgcn_ubfe(bperm((t >> 5) << 2, plane), (uint32_t)(t & 31), 1u) != 0u; };
    auto emit = [&](int p, auto&& value) {
      float* base = obs + (size_t)p * (size_t)stride;
      const int sh = (int)((a0 + (uint32_t)p * (uint32_t)stride) & 63u);
#pragma unroll
      for (int k = 0; k <= NSLOT; ++k) {
        if (64 * k - sh < stride) {                  // wave-uniform
          const int t = 64 * k - sh + lane;
          const bool ok = t >= 0 && t < stride;
          const int tt = ok ? t : 0;
          const float v = value(k, tt, sh, ok && tt < b.N);
          if (ok) st_stream<GVEC_NT_MASK>(base + t, v);
        }
      }
    };
    emit(0, [&](int, int t, int, bool in) { const bool vis = at(seen, t); return (vis && in) ? 1.0f : 0.0f; });          // :312-314
    emit(1, [&](int, int t, int, bool in) {                                                                          // :316-322 (owner -1 unless visible)
      const bool vis = at(seen, t), mine = at(own_p, t), owned = at(own_any, t);
      return (in && vis && mine) ? 0.5f : ((in && vis && owned) ? 1.0f : 0.0f);
    });
    emit(2, [&](int k, int t, int sh, bool in) {
      // the army of tile 64k - sh + l sits in slot k (lanes l >= sh) or k - 1 (l < sh), sh lanes further on
      const int from = ((lane - sh) & 63) << 2;
      const int32_t cur = (int32_t)bperm(from, (uint32_t)b.army[k < NSLOT ? k : NSLOT - 1]);
      const int32_t prv = (int32_t)bperm(from, (uint32_t)b.army[k > 0 ? k - 1 : 0]);
      const bool vis = at(seen, t);
      const int32_t army = (vis && in) ? ((lane >= sh) ? cur : prv) : 0;         // hidden and fogged tiles: army 0
      // np.log(army + 1) / 10.0 in float64, cast on store (:324-326)
      return (army > 0) ? (float)(log((double)army + 1.0) / 10.0) : 0.0f;
    });
    // (every at() is a ds_bpermute and reads zeros from lanes that sit out: never behind a lane-dependent `&&`)
    emit(3, [&](int, int t, int, bool in) {                                                                          // :328-336 one-hot type
      const bool g = at(b.gen, t), c = at(b.city, t), mt = at(b.mtn, t);
      return (in && !g && !c && !mt) ? 1.0f : 0.0f;
    });
    emit(4, [&](int, int t, int, bool in) { const bool mt = at(b.mtn, t); return (in && mt) ? 1.0f : 0.0f; });
    emit(5, [&](int, int t, int, bool in) { const bool c = at(b.city, t); return (in && c) ? 1.0f : 0.0f; });
    emit(6, [&](int, int t, int, bool in) { const bool g = at(b.gen, t); return (in && g) ? 1.0f : 0.0f; });
    emit(7, [&](int, int, int, bool) { return tc; });                                                                // the whole plane, like obs[7, :, :] = ...
    emit(8, [&](int, int, int, bool) { return 0.0f; });                                                              // left zero by the reference (:341-343)
#pragma unroll
    for (int s = 0; s < NSLOT; ++s) {
      const int t = 64 * s + lane;
      const uint32_t k0 = b.gather(m0, s), k1 = b.gather(m1, s), k2 = b.gather(m2, s), k3 = b.gather(m3, s), k4 = b.gather(many, s);
      if (t < stride) {
        uint8_t* mk = ms + t * 5;
        mk[0] = (uint8_t)k0;
        mk[1] = (uint8_t)k1;
        mk[2] = (uint8_t)k2;
        mk[3] = (uint8_t)k3;
        mk[4] = (uint8_t)k4;
      }
    }
  } else if constexpr (NSLOT >= 7) {
    // Boards of more than 256 tiles whose planes are a multiple of four floats on 16-byte boundaries (20x20, 24x25, 32x32):
    // FOUR neighbouring tiles per lane.  (65,536 envs: 20x20 0.255 -> 0.241 ms; smaller boards leave too many lanes without
    // a quad - 10x10 0.144 -> 0.159, 16x16 0.184 -> 0.195 - and keep the per-slot form below.)  Their bits are one nibble
    // of a plane's dword - one ds_bpermute per plane and 256 tiles instead of one per 64 - their armies one 16-byte LDS read,
    // each of the nine stores a 1-KB run of the wave, and the twenty mask bytes of the four tiles five dwords into the stage.
    const int nq = stride >> 2;
    constexpr int QI = (NSLOT + 3) / 4;
    int32_t* as = reinterpret_cast<int32_t*>(ms);             // the stage first carries the armies, tile t at dword t
#pragma unroll
    for (int s = 0; s < NSLOT; ++s) as[64 * s + lane] = b.army[s];
    wave_lds_fence();
    int4 a4[QI];
#pragma unroll
    for (int i = 0; i < QI; ++i) {
      const int q = lane + 64 * i;
      a4[i] = (q < nq) ? *reinterpret_cast<const int4*>(as + 4 * q) : make_int4(0, 0, 0, 0);
    }
    wave_lds_fence();                                         // ... and is free for the mask bytes from here on
#pragma unroll
    for (int i = 0; i < QI; ++i) {
      if (64 * i < nq) {                                      // wave-uniform: the bpermutes below need every lane
        const int q = lane + 64 * i;
        const bool ok = q < nq;
        const int qq = ok ? q : 0;
        const int from = (qq >> 3) << 2, sh4 = (qq & 7) << 2;
        const uint32_t n_vis = (bperm(from, seen) >> sh4) & 15u, n_mine = (bperm(from, own_p) >> sh4) & 15u;
        const uint32_t n_any = (bperm(from, own_any) >> sh4) & 15u, n_g = (bperm(from, b.gen) >> sh4) & 15u;
        const uint32_t n_c = (bperm(from, b.city) >> sh4) & 15u, n_mt = (bperm(from, b.mtn) >> sh4) & 15u;
        uint32_t kd[5];
        kd[0] = (bperm(from, m0) >> sh4) & 15u;
        kd[1] = (bperm(from, m1) >> sh4) & 15u;
        kd[2] = (bperm(from, m2) >> sh4) & 15u;
        kd[3] = (bperm(from, m3) >> sh4) & 15u;
        kd[4] = (bperm(from, many) >> sh4) & 15u;
        const int t0 = qq << 2, left = b.N - t0;              // a smaller board in a padded batch ends inside or before the quad
        const uint32_t n_in = left >= 4 ? 15u : (left <= 0 ? 0u : ((1u << left) - 1u));
        const uint32_t v = n_vis & n_in;
        if (ok) {
          const size_t n = (size_t)stride;
          auto put = [&](int p, float x, float y, float z, float w) {
            float4 o;
            o.x = x; o.y = y; o.z = z; o.w = w;
            st_stream<GVEC_NT_MASK>(reinterpret_cast<u32x4*>(obs + p * n) + q, *reinterpret_cast<const u32x4*>(&o));
          };
          auto ones = [&](int p, uint32_t m) { put(p, (m & 1u) ? 1.0f : 0.0f, (m & 2u) ? 1.0f : 0.0f, (m & 4u) ? 1.0f : 0.0f, (m & 8u) ? 1.0f : 0.0f); };
          ones(0, v);                                                                                         // :312-314
          const uint32_t mine = v & n_mine, other = v & ~n_mine & n_any;                                      // :316-322 (owner -1 unless visible)
          put(1, (mine & 1u) ? 0.5f : ((other & 1u) ? 1.0f : 0.0f), (mine & 2u) ? 0.5f : ((other & 2u) ? 1.0f : 0.0f),
              (mine & 4u) ? 0.5f : ((other & 4u) ? 1.0f : 0.0f), (mine & 8u) ? 0.5f : ((other & 8u) ? 1.0f : 0.0f));
          // channel 2: np.log(army + 1) / 10.0 in float64, cast on store (:324-326); hidden and fogged tiles: army 0
          auto la = [](bool shown, int32_t a) { return (shown && a > 0) ? (float)(log((double)a + 1.0) / 10.0) : 0.0f; };
          put(2, la((v & 1u) != 0u, a4[i].x), la((v & 2u) != 0u, a4[i].y), la((v & 4u) != 0u, a4[i].z), la((v & 8u) != 0u, a4[i].w));
          ones(3, n_in & ~n_g & ~n_c & ~n_mt);                                                                // :328-336 one-hot type
          ones(4, n_in & n_mt);
          ones(5, n_in & n_c);
          ones(6, n_in & n_g);
          put(7, tc, tc, tc, tc);                                                                             // the whole plane, like obs[7, :, :] = ...
          put(8, 0.0f, 0.0f, 0.0f, 0.0f);                                                                     // left zero by the reference (:341-343)
          // mask byte 5 * tile + d of the quad's four tiles: twenty bytes = five dwords at byte 20 * q of the stage
          uint32_t w[5] = {0u, 0u, 0u, 0u, 0u};
#pragma unroll
          for (int bb = 0; bb < 20; ++bb) w[bb >> 2] |= ((kd[bb % 5] >> (bb / 5)) & 1u) << (8 * (bb & 3));
          uint32_t* mw = reinterpret_cast<uint32_t*>(ms) + 5 * q;
#pragma unroll
          for (int k = 0; k < 5; ++k) mw[k] = w[k];
        }
      }
    }
  
  } else {
  #pragma unroll
    for (int s = 0; s < NSLOT; ++s) {
      const int t = 64 * s + lane;
      const bool vis = b.gather(seen, s) != 0u, mine = b.gather(own_p, s) != 0u, owned = b.gather(own_any, s) != 0u;
      const bool g = b.gather(b.gen, s) != 0u, c = b.gather(b.city, s) != 0u, mt = b.gather(b.mtn, s) != 0u;
      const uint32_t k0 = b.gather(m0, s), k1 = b.gather(m1, s), k2 = b.gather(m2, s), k3 = b.gather(m3, s), k4 = b.gather(many, s);
      const int32_t army = vis ? b.army[s] : 0;                                    // hidden and fogged tiles: army 0
      // channel 2: np.log(army + 1) / 10.0 in float64, cast on store (:324-326)
      const float la = (army > 0) ? (float)(log((double)army + 1.0) / 10.0) : 0.0f;
      if (t < stride) {
        const bool in = t < b.N;
        const size_t n = (size_t)stride;
        st_stream<GVEC_NT_MASK>(obs + 0 * n + t, (in && vis) ? 1.0f : 0.0f);                                // :312-314
        st_stream<GVEC_NT_MASK>(obs + 1 * n + t, (in && vis && mine) ? 0.5f : ((in && vis && owned) ? 1.0f : 0.0f));   // :316-322 (owner -1 unless visible)
        st_stream<GVEC_NT_MASK>(obs + 2 * n + t, in ? la : 0.0f);
        st_stream<GVEC_NT_MASK>(obs + 3 * n + t, (in && !g && !c && !mt) ? 1.0f : 0.0f);                    // :328-336 one-hot type
        st_stream<GVEC_NT_MASK>(obs + 4 * n + t, (in && mt) ? 1.0f : 0.0f);
        st_stream<GVEC_NT_MASK>(obs + 5 * n + t, (in && c) ? 1.0f : 0.0f);
        st_stream<GVEC_NT_MASK>(obs + 6 * n + t, (in && g) ? 1.0f : 0.0f);
        st_stream<GVEC_NT_MASK>(obs + 7 * n + t, tc);                                                       // the whole plane, like obs[7, :, :] = ...
        st_stream<GVEC_NT_MASK>(obs + 8 * n + t, 0.0f);                                                     // left zero by the reference (:341-343)
        uint8_t* mk = ms + t * 5;
        mk[0] = (uint8_t)k0;
        mk[1] = (uint8_t)k1;
        mk[2] = (uint8_t)k2;
        mk[3] = (uint8_t)k3;
        mk[4] = (uint8_t)k4;
      }
    }
  
  }
  wave_lds_fence();
  // the mask is five bytes per tile: laid out in LDS above and stored as whole 16-byte (or 4-byte) pieces of consecutive
  // lanes - five byte stores per lane and slot, each lane 5 bytes from its neighbour, held this kernel at 1.4 TB/s
  const int nbytes = 5 * stride;
  if ((nbytes & 15) == 0 && (reinterpret_cast<uintptr_t>(mask) & 15u) == 0u) {
    const u32x4* s4 = reinterpret_cast<const u32x4*>(ms);
    u32x4* g4 = reinterpret_cast<u32x4*>(mask);
    for (int i = lane; i < (nbytes >> 4); i += 64) st_stream<GVEC_NT_MASK>(g4 + i, s4[i]);
  } else if ((nbytes & 3) == 0 && (reinterpret_cast<uintptr_t>(mask) & 3u) == 0u) {
    const uint32_t* s1 = reinterpret_cast<const uint32_t*>(ms);
    uint32_t* g1 = reinterpret_cast<uint32_t*>(mask);
    for (int i = lane; i < (nbytes >> 2); i += 64) st_stream<GVEC_NT_MASK>(g1 + i, s1[i]);
  } else {
    for (int i = lane; i < nbytes; i += 64) mask[i] = ms[i];
  }
}

// _calculate_reward (generals_env.py:499-561) against the stats the previous call stored, GeneralEnv.step's bookkeeping
// around it (:226-259) when `flow`, then the new stats.  cur_tc / cur_ac: the learner's tile_count (= len(OwnedTiles),
// server.go:536) and ArmyCount; tcl / acl: lane p < MAXP holds player p's.  Every lane calls it.
struct GymFlowOut {
  double* reward;
  uint8_t* done;
  int8_t* winner;
  int64_t* turn_io;
  int64_t* turn_out;
  uint8_t* terminated;
  uint8_t* truncated;
  uint8_t* needs_reset;
};
template <int MAXP>
__device__ __forceinline__ void gym_bookkeeping(int env, int pl, int P, uint32_t alive, bool over, int32_t cur_tc, int32_t cur_ac, uint32_t tcl,
                                                uint32_t acl, int32_t* prev, bool flow, bool rs, bool pl_ok, int64_t turns, int max_turns,
                                                const GymFlowOut& O) {
  const int lane = lane_id();
  const int na = __builtin_popcount(alive);
  const int winner = (over && P > 1 && na == 1) ? (31 - __builtin_clz(alive)) : -1;   // Engine.GetWinner
  if (lane == 0) {
    double r = 0.0;
    r += (double)(cur_tc - prev[pl]) * 1.0;                                       // :540-542
    r += (double)(cur_ac - prev[MAXP + pl]) * 0.01;                               // :544-546
    for (int q = 0; q < P; ++q)                                                   // :548-555
      if (q != pl && prev[2 * MAXP + q] != 0 && !((alive >> q) & 1u)) r += 50.0;
    if (over) r = (winner == pl) ? 100.0 : -100.0;                                // :520-524
    if (flow) {
      // :226-241 a refused action costs -0.1 and changes nothing; :243-259 terminated = game over, truncated = turn limit
      const bool term = over && pl_ok && !rs, trunc = turns >= (int64_t)max_turns && pl_ok && !rs;
      if (O.reward) O.reward[env] = rs ? 0.0 : (pl_ok ? r : -0.1);
      if (O.winner) O.winner[env] = (int8_t)(term ? winner : -1);
      O.turn_io[env] = turns;
      if (O.turn_out) O.turn_out[env] = turns;
      if (O.terminated) O.terminated[env] = (uint8_t)(term ? 1 : 0);
      if (O.truncated) O.truncated[env] = (uint8_t)(trunc ? 1 : 0);
      if (O.needs_reset) O.needs_reset[env] = (uint8_t)((term || trunc) ? 1 : 0);
    } else {
      if (O.reward) O.reward[env] = r;
      if (O.winner) O.winner[env] = (int8_t)winner;
    }
    if (O.done) O.done[env] = (uint8_t)(over ? 1 : 0);
  }
  if (lane < MAXP) {
    prev[lane] = (int32_t)tcl;
    prev[MAXP + lane] = (int32_t)acl;
    prev[2 * MAXP + lane] = (int32_t)((alive >> lane) & 1u);
  }
}

template <int MAXP, int NSLOT>
__global__ __launch_bounds__(64 * WAVES_PER_BLOCK) void gym_observe_kernel(GymArgs A) {
  using B = Board<MAXP, NSLOT>;
  const int wave = (int)(threadIdx.x >> 6), lane = lane_id();
  const int env = uni((int)blockIdx.x * WAVES_PER_BLOCK + wave);
  if (env >= A.num_envs) return;
  B b;
  load_board(b, A.hdr + (size_t)env * HDR_DW, A.rows + (size_t)env * A.row_dw, army_cref<NSLOT>(A.army16, A.army32, env), A.fd);
  const bool fog_on = (b.hflags & HF_FOG) != 0u;
  uint32_t own_p = 0u, vis_p = 0u, own_any = 0u, lst_cnt[MAXP];
#pragma unroll
  for (int p = 0; p < MAXP; ++p) {
    own_p = (p == A.player) ? b.own[p] : own_p;
    vis_p = (p == A.player) ? b.vis[p] : vis_p;
    own_any |= b.own[p];
    lst_cnt[p] = (uint32_t)b.count(b.lst[p]);  // PlayerState.tile_count = len(OwnedTiles) (server.go:536)
  }
  const uint32_t seen = fog_on ? vis_p : b.valid;  // ComputePlayerVisibility (visibility_optimized.go:166-195)
  // _get_valid_actions_mask: a tile the proto shows as ours (visible, owner == player) with army > 1, towards a
  // neighbour on the board whose shown type is not MOUNTAIN; index tile*5 + {up, right, down, left}, +4 = half move
  const uint32_t src = own_p & seen & b.gt1;
  const uint32_t m0 = src & b.ok[0], m1 = src & b.ok[1], m2 = src & b.ok[2], m3 = src & b.ok[3], many = m0 | m1 | m2 | m3;
  __shared__ uint32_t mask_stage[WAVES_PER_BLOCK][(NSLOT * 64 * 5 + 15) / 16 * 4];
  // GeneralsEnv.step's bookkeeping around the observation (generals_env.py:226-259), when asked for: a re-dealt env
  // restarts its turn count, a refused action leaves it alone
  const bool flow = A.played != nullptr;
  const bool rs = flow && A.resetting[env] != 0, pl_ok = !flow || A.played[env] != 0;
  const int64_t turns = flow ? (rs ? 0 : A.turn_count[env] + (pl_ok ? 1 : 0)) : A.turn_count[env];
  // channel 7: min(turn_count / max_turns, 1.0) in float64, stored as float32 (:338-339)
  double tcn = (double)turns / (double)A.max_turns;
  tcn = tcn < 1.0 ? tcn : 1.0;
  gym_emit<NSLOT>(b, seen, own_p, own_any, m0, m1, m2, m3, many, (float)tcn, A.obs + (size_t)env * 9 * (size_t)A.stride,
                  A.mask + (size_t)env * 5 * (size_t)A.stride, reinterpret_cast<uint8_t*>(mask_stage[wave]), A.stride);
  int32_t cur_tc = 0, cur_ac = 0;
  uint32_t tcl = 0u;
#pragma unroll
  for (int p = 0; p < MAXP; ++p) {
    cur_tc = (p == A.player) ? (int32_t)lst_cnt[p] : cur_tc;
    tcl = (lane == p) ? lst_cnt[p] : tcl;
  }
  // lanes H_ARMYCNT + p of the header register hold ArmyCount[p]: fetched by every lane (a cross-lane read must
  // not sit under a divergent branch: masked-off source lanes read as 0)
  const uint32_t acl = bperm((H_ARMYCNT + (lane & (MAXP - 1))) << 2, b.hv);
  cur_ac = (int32_t)rdlane(acl, A.player);
  const GymFlowOut O{A.reward, A.done, A.winner, A.turn_io, A.turn_out, A.terminated, A.truncated, A.needs_reset};
  gym_bookkeeping<MAXP>(env, A.player, b.P, b.alive, (b.hflags & HF_DONE) != 0u, cur_tc, cur_ac, tcl, acl,
                        A.prev_stats + (size_t)env * 3 * MAXP, flow, rs, pl_ok, turns, A.max_turns, O);
}

// GeneralsEnv.step for every env in ONE launch (gvec_gym_step) = gvec_agent_actions + gvec_gym_actions + gvec_step +
// gvec_gym_finish_step, which it equals bit for bit (tests/test_vector_env.py): the learner's Discrete(N*5) action is
// decoded against the valid-action mask of the resident state (recomputed from the planes in registers - the bytes
// gym_observe wrote are not read back), the opponents' moves come from the on-device agent, the turn is played, and the
// observation / mask / reward / flags of the NEW state leave while the board is still in registers.
// (Five waves per SIMD asked for by name: left alone the compiler takes 107-145 VGPRs - four waves, three for the largest
// boards; told to fit five it needs 81-96 and spills nothing except 24-28 bytes in <8,16>.  65,536 envs: 16x16 0.201 ->
// 0.184 ms, 20x20 4P 0.290 -> 0.260, 10x10 0.157 -> 0.140, 32x32 8P 0.586 -> 0.556, 15x15 and 25x25 unchanged; six waves
// (73-80 VGPRs) gain on boards of up to 256 tiles and lose 8-15 % on every larger one.)
template <int MAXP, int NSLOT, bool ODD>
__global__ __launch_bounds__(64 * WAVES_PER_BLOCK) __attribute__((amdgpu_waves_per_eu(5, 5))) void gym_step_kernel(StepArgs A, GymStepArgs G) {
  constexpr int FD = 2 * NSLOT - (ODD ? 1 : 0);
  constexpr int ROW_DW = (Planes<MAXP>::COUNT * FD + 3) / 4 * 4;
  constexpr int STAGE_DW = (NSLOT * 64 * 5 + 15) / 16 * 4;  // the gym mask's stage (5 bytes a tile) is the larger user of the army shadow
  static_assert(STAGE_DW >= NSLOT * 64, "the stage also serves as the action phase's army shadow");
  using B = Turn<MAXP, NSLOT>;
  constexpr int PPR = B::PPR, ROWL = B::ROWL, NR = B::NR;
  __shared__ int32_t army_shadow[WAVES_PER_BLOCK][STAGE_DW];
  __shared__ uint32_t act_scratch[WAVES_PER_BLOCK][B::ACT_SCRATCH_DW];
  const int wave = (int)(threadIdx.x >> 6), lane = lane_id();
  const int env = uni((int)blockIdx.x * WAVES_PER_BLOCK + wave);
  if (env >= A.num_envs) return;
  B b;
  b.larmy = army_shadow[wave];
  b.lscr = act_scratch[wave];
  const ArmyRef army_env = army_ref<NSLOT>(A.army16, A.army32, env);
  load_turn<true>(b, A.hdr + (size_t)env * HDR_DW, A.rows + (size_t)env * ROW_DW, army_env, FD, A.zeros);
  b.small = !(b.hflags & HF_WIDE);
  const int pl = G.player;
  // the learner's planes, replicated into every row: row pl % PPR of register pl / PPR
  auto learner = [&](const uint32_t (&reg)[NR]) {
    uint32_t out = 0u;
#pragma unroll
    for (int k = 0; k < NR; ++k) {
      const uint32_t g = bperm((((pl % PPR) * ROWL) + B::col()) << 2, reg[k]);
      out = (pl / PPR == k) ? g : out;
    }
    return out;
  };
  // ---- GeneralsEnv.step's action handling (generals_env.py:226-259, :389-441; gym_actions_kernel) -------------
  const long long a = (long long)uni64((uint64_t)G.gym_actions[env]);
  const bool rs = uni((int)G.resetting[env]) != 0;
  bool valid, accepted, half;
  int from, d;
  {
    const uint32_t seen = (b.hflags & HF_FOG) ? learner(b.vis) : b.valid;
    const uint32_t src = learner(b.own) & seen & b.gt1;
    const long long n5 = 5ll * G.stride;
    const bool in_range = a >= 0 && a < n5;
    from = in_range ? (int)(a / 5) : 0;
    const int info = in_range ? (int)(a % 5) : 0;
    const int fy = (int)(__umul24((uint32_t)from, (uint32_t)b.recipW) >> 16), fx = from - (int)__umul24((uint32_t)fy, (uint32_t)b.W);  // from < 1024
    half = info == 4;
    d = half ? 3 : info;
    if (half) {  // the FIRST of up / right / down / left whose target is on the board (mountains are not checked there)
      if (fx - 1 >= 0) d = 3;
      if (fy + 1 < b.H) d = 2;
      if (fx + 1 < b.W) d = 1;
      if (fy - 1 >= 0) d = 0;
    }
    // a tile index beyond the env's own board has no mask bit: the planes are zero there
    const uint32_t wsrc = rdlane(src, from >> 5);
    const uint32_t o0 = rdlane(b.ok[0], from >> 5), o1 = rdlane(b.ok[1], from >> 5), o2 = rdlane(b.ok[2], from >> 5), o3 = rdlane(b.ok[3], from >> 5);
    const uint32_t bit = 1u << (from & 31);
    const bool s_ok = (wsrc & bit) != 0u;
    const bool k0 = s_ok && (o0 & bit), k1 = s_ok && (o1 & bit), k2 = s_ok && (o2 & bit), k3 = s_ok && (o3 & bit);
    const bool kinfo = (info == 0) ? k0 : (info == 1) ? k1 : (info == 2) ? k2 : (info == 3) ? k3 : (k0 || k1 || k2 || k3);
    valid = in_range && kinfo;
    const bool kd = (d == 0) ? k0 : (d == 1) ? k1 : (d == 2) ? k2 : k3;
    accepted = valid && kd;   // the server validates the move it received (action_validator.go:114-139)
  }
  const bool played = accepted || rs;
  if (lane == 0) {
    if (G.played) G.played[env] = (uint8_t)played;
    if (G.invalid) G.invalid[env] = (uint8_t)(!valid && !rs);
    if (G.error) G.error[env] = (uint8_t)(valid && !accepted && !rs);
  }
  // ---- the turn (step_kernel's body; an env whose action was refused sits the call out) --------------------------
  if (played) {
    uint32_t err = 0u;
    bool types_dirty = false;
    if ((b.hflags & HF_DONE) || rs) {
      redeal<MAXP, NSLOT>(b, A, env, FD, ROW_DW);
      types_dirty = true;
    } else {
      uint32_t m[NR][4];
      b.template legal_planes<false>(m);
      const uint32_t mine = agent_sample<MAXP, NSLOT>(b, m, env_key_of(A.seed_base, (uint32_t)env), A);
      typename B::ActVec av = agent_actvec<MAXP, NSLOT>(b, mine, A.invalid_permille > 0);
      // the learner's slot: the accepted move (on the board and legal by construction of the mask)
      const int tt = from + ((d == 0) ? -b.W : (d == 1) ? 1 : (d == 2) ? b.W : -1);
      av.meta = (lane == pl) ? (16u | (half ? 32u : 0u)) : av.meta;
      av.ft = (lane == pl) ? from : av.ft;
      av.tt = (lane == pl) ? tt : av.tt;
      bool aborted;
      err = b.turn_step(av, A, aborted);
      b.refresh_gt1();
      b.hdr_set(H_CNT_STEPS, b.hdr_get(H_CNT_STEPS) + 1u);
      if (aborted) b.hdr_set(H_CNT_ABORT, b.hdr_get(H_CNT_ABORT) + 1u);
      if (b.hflags & HF_DONE) b.hdr_set(H_CNT_DONE, b.hdr_get(H_CNT_DONE) + 1u);
    }
    b.store_army_staged(army_env);
    b.settle_lists();
    b.store_hdr(A.hdr + (size_t)env * HDR_DW, err);
    if (types_dirty) b.store_planes(A.rows + (size_t)env * ROW_DW, FD, ROW_DW, true);
    else b.store_planes_staged(A.rows + (size_t)env * ROW_DW, FD);
    if (A.err && lane == 0) A.err[env] = (int32_t)err;
  } else if (A.err && lane == 0) {
    A.err[env] = 0;
  }
  // ---- observation, mask, reward, flags of the state as it is now (gym_observe_kernel's body) ----------------------
  const uint32_t own_p = learner(b.own);
  const uint32_t seen = (b.hflags & HF_FOG) ? learner(b.vis) : b.valid;
  uint32_t own_any = 0u;
#pragma unroll
  for (int k = 0; k < NR; ++k) own_any |= b.own[k];
  own_any = B::or_rows(own_any);
  const uint32_t src = own_p & seen & b.gt1;
  const uint32_t m0 = src & b.ok[0], m1 = src & b.ok[1], m2 = src & b.ok[2], m3 = src & b.ok[3], many = m0 | m1 | m2 | m3;
  const int64_t turns = rs ? 0 : G.turn_io[env] + (played ? 1 : 0);
  double tcn = (double)turns / (double)G.max_turns;
  tcn = tcn < 1.0 ? tcn : 1.0;
  wave_lds_fence();  // the staged state stores above have read the stage
  gym_emit<NSLOT>(b, seen, own_p, own_any, m0, m1, m2, m3, many, (float)tcn, G.obs + (size_t)env * 9 * (size_t)G.stride,
                  G.mask + (size_t)env * 5 * (size_t)G.stride, reinterpret_cast<uint8_t*>(army_shadow[wave]), G.stride);
  // per-player len(OwnedTiles): row totals in the rows' last lanes, handed to lane p
  uint32_t tcl = 0u;
#pragma unroll
  for (int k = 0; k < NR; ++k) {
    const uint32_t sc = row_scan_add<ROWL>((uint32_t)__builtin_popcount(b.lst[k]));
    const uint32_t got = bperm((((lane % PPR) * ROWL) + ROWL - 1) << 2, sc);
    tcl = (lane / PPR == k) ? got : tcl;
  }
  const uint32_t acl = bperm((H_ARMYCNT + (lane & (MAXP - 1))) << 2, b.hv);
  const GymFlowOut O{G.reward, nullptr, G.winner, G.turn_io, G.turn_out, G.terminated, G.truncated, G.needs_reset};
  gym_bookkeeping<MAXP>(env, pl, b.P, b.alive, (b.hflags & HF_DONE) != 0u, (int32_t)rdlane(tcl, pl), (int32_t)rdlane(acl, pl), tcl, acl,
                        G.prev_stats + (size_t)env * 3 * MAXP, true, rs, played, turns, G.max_turns, O);
}

// GeneralsEnv.step's action handling for player `player` of every env (one thread per env):
// :226-241 an action the mask rejects is not submitted (the env sits the call out: GVEC_ACT_SKIP_ENV);
// _action_index_to_game_action :389-441 (a half move, index 4, takes the FIRST of up / right / down / left whose
// target is on the board - mountains are not checked there); the server then validates the move it received
// (action_validator.go:114-139): a half move whose first in-board direction is illegal is refused.
// `resetting` envs are re-dealt in this step (GVEC_ACT_RESET_ENV) whatever the action.
__global__ void gym_actions_kernel(GymActArgs A) {
  const int env = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (env >= A.num_envs) return;
  const uint32_t dims = A.hdr[(size_t)env * HDR_DW + H_DIMS];
  const int w = (int)(dims & 0xFFu), h = (int)((dims >> 8) & 0xFFu);
  const long long a = A.gym_actions[env];
  const long long n5 = 5ll * A.stride;
  const uint8_t* mask = A.mask + (size_t)env * 5 * (size_t)A.stride;
  const bool in_range = a >= 0 && a < n5;
  const bool valid = in_range && mask[a] != 0;
  const int from = in_range ? (int)(a / 5) : 0, info = in_range ? (int)(a % 5) : 0;
  const int fx = from % w, fy = from / w;   // tile index with the env's own width (from < stride; a tile beyond the board has no mask bit)
  const bool half = info == 4;
  int d = half ? 3 : info;
  if (half) {
    if (fx - 1 >= 0) d = 3;
    if (fy + 1 < h) d = 2;
    if (fx + 1 < w) d = 1;
    if (fy - 1 >= 0) d = 0;
  }
  const bool accepted = valid && mask[(size_t)from * 5 + d] != 0;
  const bool resetting = A.resetting && A.resetting[env] != 0;
  const bool played = accepted || resetting;
  const int dx = (d == 1) - (d == 3), dy = (d == 2) - (d == 0);
  gvec_action* acts = A.actions + (size_t)env * A.pstride;
  gvec_action mine;
  mine.from_x = (int8_t)fx;
  mine.from_y = (int8_t)fy;
  mine.to_x = (int8_t)(fx + dx);
  mine.to_y = (int8_t)(fy + dy);
  mine.flags = (uint8_t)(played ? (GVEC_ACT_VALID | (half ? GVEC_ACT_HALF : 0u)) : 0u);
  mine.reserved[0] = mine.reserved[1] = mine.reserved[2] = 0;
  acts[A.player] = mine;
  uint8_t f0 = acts[0].flags & (uint8_t)~(GVEC_ACT_SKIP_ENV | GVEC_ACT_RESET_ENV);
  if (!played) f0 |= GVEC_ACT_SKIP_ENV;
  if (resetting) f0 |= GVEC_ACT_RESET_ENV;
  acts[0].flags = f0;
  if (A.played) A.played[env] = (uint8_t)played;
  if (A.invalid) A.invalid[env] = (uint8_t)(!valid && !resetting);
  if (A.error) A.error[env] = (uint8_t)(valid && !accepted && !resetting);
}

// =========================================================================================
// gameInstance.createStreamUpdate's delta (internal/grpc/gameserver/server.go:636-777) for one player's stream, every env:
// when 0 < |ChangedTiles| + |VisibilityChangedTiles| < N / 5 (a tile in both sets counts twice, :636-640) the update is a
// GameStateDelta whose tile updates are the tiles of either set with the proto's fog rules applied for that player
// (:664-689 == :556-582); otherwise the server sends the full state.  The handful of tiles a turn touches leave the GPU
// instead of the board: ~10 eight-byte updates per env instead of 4 KB of planes.
// updates[env][k] = tile index | type << 16 | visible << 18 | fog_of_war << 19 | (owner + 1) << 20 | army << 32, the changed
// tiles ascending, then the visibility-only ones ascending (Go ranges over maps: its order is unspecified).
// =========================================================================================
template <int MAXP, int NSLOT>
__global__ __launch_bounds__(64 * WAVES_PER_BLOCK) void stream_delta_kernel(StreamDeltaArgs A) {
  using B = Board<MAXP, NSLOT>;
  const int wave = (int)(threadIdx.x >> 6), lane = lane_id();
  const int env = uni((int)blockIdx.x * WAVES_PER_BLOCK + wave);
  if (env >= A.num_envs) return;
  B b;
  load_board(b, A.hdr + (size_t)env * HDR_DW, A.rows + (size_t)env * A.row_dw, army_cref<NSLOT>(A.army16, A.army32, env), A.fd);
  const int nc = b.count(b.chg), nv = b.count(b.vch);
  const int total = nc + nv;
  const bool delta = total > 0 && total < b.N / 5;       // :640 (integer division)
  const int n_union = b.count(b.chg | b.vch);            // a wave-wide reduction: every lane takes part
  const bool all_tiles = !delta && A.full_tiles != 0;     // the full state's tiles (server.go:556-582) for envs that get no delta
  if (lane == 0) {
    A.kind[env] = (uint8_t)(delta ? 1 : 2);
    A.count[env] = delta ? n_union : (all_tiles ? b.N : 0);
  }
  if (!delta && !all_tiles) return;                       // wave-uniform
  const bool fog_on = (b.hflags & HF_FOG) != 0u;
  uint32_t vis_p = 0u;
#pragma unroll
  for (int p = 0; p < MAXP; ++p) vis_p = (p == A.player) ? b.vis[p] : vis_p;
  unsigned long long* out = A.updates + (size_t)env * A.cap;
  int base = 0;
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const uint32_t sel_plane = all_tiles ? (pass == 0 ? b.valid : 0u) : (pass == 0 ? b.chg : (b.vch & ~b.chg));
#pragma unroll
    for (int s = 0; s < NSLOT; ++s) {
      const int t = 64 * s + lane;
      int owner = -1;
#pragma unroll
      for (int p = 0; p < MAXP; ++p) owner = b.gather(b.own[p], s) ? p : owner;
      const uint32_t is_gen = b.gather(b.gen, s), is_city = b.gather(b.city, s), is_mtn = b.gather(b.mtn, s);
      const uint32_t pv = b.gather(vis_p, s);
      const bool sel = b.gather(sel_plane, s) != 0u && t < b.N;
      int type = is_gen ? GVEC_TILE_GENERAL : (is_city ? GVEC_TILE_CITY : (is_mtn ? GVEC_TILE_MOUNTAIN : GVEC_TILE_NORMAL));
      const bool visible = !fog_on || pv != 0u;                        // ComputePlayerVisibility (visibility_optimized.go:166-195)
      const bool fogged = !visible && type != GVEC_TILE_NORMAL;
      int32_t army = b.army[s];
      if (!visible) {                                                  // :676-688: hidden or fogged - the current state is withheld
        owner = -1;
        army = 0;
      }                                                                // (a hidden tile IS a normal tile: its type needs no rewrite)
      const unsigned long long m = __builtin_amdgcn_ballot_w64(sel);
      const int pos = base + __builtin_popcountll(m & ((1ull << lane) - 1ull));
      if (sel && pos < A.cap)
        out[pos] = (unsigned long long)((uint32_t)t | ((uint32_t)type << 16) | ((visible ? 1u : 0u) << 18) | ((fogged ? 1u : 0u) << 19) |
                                        ((uint32_t)(owner + 1) << 20)) |
                   ((unsigned long long)(uint32_t)army << 32);
      base += __builtin_popcountll(m);
    }
  }
}

// gvec_stream_deltas_packed: exclusive prefix sum of the per-env update counts (one workgroup: B is a few hundred thousand
// small integers) and the row-to-stream compaction that follows it.
__global__ __launch_bounds__(1024) void scan_counts_kernel(const int32_t* count, long long* offset, int32_t n) {
  __shared__ long long part[1024];
  const int tid = (int)threadIdx.x;
  const int per = (n + 1023) / 1024;
  const int lo = tid * per, hi = (lo + per < n) ? lo + per : n;
  long long sum = 0;
  for (int i = lo; i < hi; ++i) sum += count[i];
  part[tid] = sum;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {   // Hillis-Steele inclusive scan of the 1,024 partial sums
    const long long add = (tid >= off) ? part[tid - off] : 0;
    __syncthreads();
    part[tid] += add;
    __syncthreads();
  }
  long long run = part[tid] - sum;             // exclusive base of this thread's chunk
  for (int i = lo; i < hi; ++i) {
    offset[i] = run;
    run += count[i];
  }
  if (tid == 1023) offset[n] = part[1023];
}
__global__ __launch_bounds__(64 * WAVES_PER_BLOCK) void pack_updates_kernel(const unsigned long long* rows, const int32_t* count, const long long* offset,
                                                                            unsigned long long* packed, int32_t n, int32_t cap, long long capacity) {
  const int env = (int)(blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6));
  if (env >= n) return;
  const long long base = offset[env];
  const int c = count[env];
  for (int k = lane_id(); k < c; k += 64)
    if (base + k < capacity) packed[base + k] = rows[(size_t)env * cap + k];
}

// =========================================================================================
// pool collection: the loop ParallelEnvPool's workers run around GeneralsEnv.step (python/generals_gym/vector_env.py:164-192)
// and ReplayBuffer.push (replay_buffer.py:31-36), for every worker at once and without leaving the device
// =========================================================================================
// scratch: flag[B] (bit 0 live: this step was a transition of the worker's episode; bit 1 over: the episode ended with it),
// per 64-worker group the exclusive prefix counts of both bits, the cursors the push works from, what finished episodes report
struct CollectScratch {
  uint8_t* flag;          // [G * 64]
  int32_t* fin_length;    // [B]
  long long* base_live;   // [G + 1]
  long long* base_over;   // [G + 1]
  long long* snap;        // [2]: the ring's cursor and the results held BEFORE this call
  double* fin_reward;     // [B]
};
__host__ __device__ inline int collect_groups(int32_t n) { return (n + 63) >> 6; }
__host__ __device__ inline CollectScratch collect_scratch(void* base, int32_t n) {
  const int g = collect_groups(n);
  CollectScratch c;
  c.base_live = static_cast<long long*>(base);
  c.base_over = c.base_live + g + 1;
  c.snap = c.base_over + g + 1;
  c.flag = reinterpret_cast<uint8_t*>(c.snap + 2);              // 16 * (g + 2) bytes in: read sixteen bytes at a time
  c.fin_reward = reinterpret_cast<double*>(c.flag + (size_t)g * 64);
  c.fin_length = reinterpret_cast<int32_t*>(c.fin_reward + n);
  return c;
}
// one thread per worker: vector_env.py:172-192 without the push
__global__ __launch_bounds__(256) void collect_flags_kernel(gvec_collect_args A) {
  const int w = (int)(blockIdx.x * 256 + threadIdx.x);
  const CollectScratch S = collect_scratch(A.scratch, A.num_envs);
  if (w >= collect_groups(A.num_envs) * 64) return;
  if (w >= A.num_envs) {
    S.flag[w] = 0;          // the tail of the last group
    return;
  }
  const bool live = !A.was_reset[w];
  const bool done = (A.terminated[w] | A.truncated[w]) != 0;
  double er = A.episode_reward[w];
  long long el = A.episode_length[w];
  if (live) {
    er += A.reward[w];      // :186
    el += 1;                // :187
  }
  const bool over = live && (done || el >= A.max_steps_per_episode);   // the while condition of :177 failing
  S.flag[w] = (uint8_t)((live ? 1 : 0) | (over ? 2 : 0));
  if (over) {
    S.fin_reward[w] = er;
    S.fin_length[w] = (int32_t)el;
    er = 0.0;
    el = 0;
    if (!done && A.needs_reset) A.needs_reset[w] = 1;   // cut at the length limit: the next step is the worker's env.reset()
  }
  A.episode_reward[w] = er;
  A.episode_length[w] = el;
}
// one workgroup: exclusive prefix counts per 64-worker group (a thread owns a run of consecutive groups, 64 flag bytes each),
// then the counters move on - the push works from the snapshot
__global__ __launch_bounds__(1024) void collect_scan_kernel(gvec_collect_args A) {
  __shared__ long long part[2][1024];
  const CollectScratch S = collect_scratch(A.scratch, A.num_envs);
  const int G = collect_groups(A.num_envs);
  const int tid = (int)threadIdx.x;
  const int per = (G + 1023) / 1024;
  const int lo = tid * per < G ? tid * per : G, hi = (lo + per < G) ? lo + per : G;
  long long nl = 0, no = 0;
  for (int g = lo; g < hi; ++g) {
    const uint4* f = reinterpret_cast<const uint4*>(S.flag + (size_t)g * 64);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const uint4 v = f[q];
      nl += __popc(v.x & 0x01010101u) + __popc(v.y & 0x01010101u) + __popc(v.z & 0x01010101u) + __popc(v.w & 0x01010101u);
      no += __popc(v.x & 0x02020202u) + __popc(v.y & 0x02020202u) + __popc(v.z & 0x02020202u) + __popc(v.w & 0x02020202u);
    }
  }
  part[0][tid] = nl;
  part[1][tid] = no;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {   // Hillis-Steele inclusive scan of the 1,024 partial sums, both counts at once
    const long long a0 = (tid >= off) ? part[0][tid - off] : 0, a1 = (tid >= off) ? part[1][tid - off] : 0;
    __syncthreads();
    part[0][tid] += a0;
    part[1][tid] += a1;
    __syncthreads();
  }
  long long rl = part[0][tid] - nl, ro = part[1][tid] - no;
  for (int g = lo; g < hi; ++g) {
    S.base_live[g] = rl;
    S.base_over[g] = ro;
    const uint4* f = reinterpret_cast<const uint4*>(S.flag + (size_t)g * 64);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const uint4 v = f[q];
      rl += __popc(v.x & 0x01010101u) + __popc(v.y & 0x01010101u) + __popc(v.z & 0x01010101u) + __popc(v.w & 0x01010101u);
      ro += __popc(v.x & 0x02020202u) + __popc(v.y & 0x02020202u) + __popc(v.z & 0x02020202u) + __popc(v.w & 0x02020202u);
    }
  }
  if (tid == 1023) {
    const long long pushed = part[0][1023], ended = part[1][1023];
    long long* R = reinterpret_cast<long long*>(A.ring_counters);
    long long* P = reinterpret_cast<long long*>(A.pool_counters);
    S.snap[0] = R[0];
    S.snap[1] = P[1];
    R[0] = (R[0] + pushed) % A.capacity;
    R[1] = (R[1] + pushed < A.capacity) ? R[1] + pushed : A.capacity;
    R[2] += pushed;
    P[0] += ended;
    const long long room = A.result_capacity - P[1];
    const long long kept = ended < room ? ended : room;
    P[1] += kept;
    P[2] += ended - kept;
  }
}
// A row of n floats from s to d, both only dword-aligned (a row is 9*W*H floats) and not alike: sixteen bytes per lane with
// BOTH the loads and the stores on 16-byte boundaries - a destination quad is cut out of two neighbouring source quads (the
// second load hits the lines the neighbouring lane fetches) - because either side misaligned costs a quarter of the rate
// (4.0-4.2 TB/s against 5.3 on this copy).  The few floats before / after the aligned body go one by one.  `part` of
// 1 << shift wavefronts share the row.
template <int D>
static __device__ __forceinline__ void copy_quads(const float4* __restrict__ sq, float4* __restrict__ dq, int jlo, int jhi, int c, int first, int stride) {
#pragma unroll 4
  for (int j = jlo + first; j < jhi; j += stride) {
    const float4 lo = sq[j + c];
    float4 o;
    if (D == 0) {
      o = lo;
    } else {
      const float4 hi = sq[j + c + 1];
      if (D == 1) o = make_float4(lo.y, lo.z, lo.w, hi.x);
      if (D == 2) o = make_float4(lo.z, lo.w, hi.x, hi.y);
      if (D == 3) o = make_float4(lo.w, hi.x, hi.y, hi.z);
    }
    dq[j] = o;
  }
}
static __device__ __forceinline__ void copy_row(const float* __restrict__ s, float* __restrict__ d, int n, int part, int lane, int shift) {
  const int ks = (int)(((16u - (unsigned)(reinterpret_cast<uintptr_t>(s) & 15u)) & 15u) >> 2);   // floats before s is 16-byte aligned
  const int kd = (int)(((16u - (unsigned)(reinterpret_cast<uintptr_t>(d) & 15u)) & 15u) >> 2);
  const int delta = uni((kd - ks) & 3), c = kd >= ks ? 0 : -1;
  // destination quad j = floats [kd + 4j, kd + 4j + 4) = source quads j + c and j + c + 1 (counted from s + ks); all of it inside the row:
  const int jlo = -c;
  const int jhi = (n - 8 - ks < 0) ? jlo : (n - 8 - ks) / 4 - c + 1;          // exclusive
  const float4* sq = reinterpret_cast<const float4*>(s + ks);
  float4* dq = reinterpret_cast<float4*>(d + kd);
  const int first = part * 64 + lane, stride = 64 << shift;
  switch (delta) {
    case 0: copy_quads<0>(sq, dq, jlo, jhi, c, first, stride); break;
    case 1: copy_quads<1>(sq, dq, jlo, jhi, c, first, stride); break;
    case 2: copy_quads<2>(sq, dq, jlo, jhi, c, first, stride); break;
    default: copy_quads<3>(sq, dq, jlo, jhi, c, first, stride); break;
  }
  if (part == 0) {
    const int head = kd + 4 * jlo, tail = kd + 4 * jhi;       // [0, head) and [tail, n): fewer than 16 floats each
    if (lane < head && lane < n) d[lane] = s[lane];
    if (tail + lane < n) d[tail + lane] = s[tail + lane];
  }
}
// `wpe` wavefronts per worker (a power of two): ReplayBuffer.push of its transition, and its episode result
__global__ __launch_bounds__(64 * WAVES_PER_BLOCK) void collect_push_kernel(gvec_collect_args A, int wpe_shift) {
  const int gw = uni((int)(blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6)));
  const int w = gw >> wpe_shift, part = gw & ((1 << wpe_shift) - 1);
  if (w >= A.num_envs) return;
  const CollectScratch S = collect_scratch(A.scratch, A.num_envs);
  const int lane = lane_id();
  const int g = w >> 6, at = w & 63;
  const uint32_t mine = S.flag[(size_t)g * 64 + lane];
  const unsigned long long below = (1ull << at) - 1;
  const unsigned long long live_m = __ballot(mine & 1), over_m = __ballot(mine & 2);
  if ((live_m >> at) & 1) {
    long long slot = S.snap[0] + S.base_live[g] + __popcll(live_m & below);
    if (slot >= A.capacity) slot -= A.capacity;          // cursor < capacity and fewer than num_envs <= capacity ahead of it
    const float* s0 = A.state + (size_t)w * A.obs_floats;
    const float* s1 = A.next_state + (size_t)w * A.obs_floats;
    float* d0 = A.ring_state + (size_t)slot * A.obs_floats;
    float* d1 = A.ring_next_state + (size_t)slot * A.obs_floats;
    copy_row(s0, d0, A.obs_floats, part, lane, wpe_shift);
    copy_row(s1, d1, A.obs_floats, part, lane, wpe_shift);
    if (part == 0) {
      if (lane == 0) {
        A.ring_action[slot] = A.action[w];
        A.ring_reward[slot] = A.reward[w];
        A.ring_done[slot] = (A.terminated[w] | A.truncated[w]) != 0;
      }
    }
  }
  if (part == 0 && lane == 0 && ((over_m >> at) & 1)) {
    const long long j = S.snap[1] + S.base_over[g] + __popcll(over_m & below);
    if (j < A.result_capacity) {
      A.result_reward[j] = S.fin_reward[w];
      A.result_length[j] = S.fin_length[w];
      A.result_worker[j] = w;
    }
  }
}

// =========================================================================================
// import: planes -> resident record (gvec_reset / gvec_write_state / pool build)
// =========================================================================================
template <int MAXP, int NSLOT>
__global__ __launch_bounds__(64 * WAVES_PER_BLOCK) void import_kernel(ImportArgs A) {
  using B = Board<MAXP, NSLOT>;
  const int wave = (int)(threadIdx.x >> 6), lane = lane_id();
  const int i = uni((int)blockIdx.x * WAVES_PER_BLOCK + wave);
  if (i >= A.n) return;
  const int env = A.env_ids ? uni(A.env_ids[i]) : A.dst_begin + i;
  if (env < 0 || env >= A.dst_envs) {  // ids handed over in device memory were not seen by the host
    if (lane == 0) atomicExch(A.status, GVEC_E_RANGE);
    return;
  }
  uint32_t* hdr = A.hdr + (size_t)env * HDR_DW;
  uint32_t* rows = A.rows + (size_t)env * A.row_dw;
  const ArmyRef army = army_ref<NSLOT>(A.army16, A.army32, env);
  const size_t to = (size_t)i * A.stride, po = (size_t)i * A.max_p;

  B b;
  if (A.fresh) {
    b.W = A.s_width[i];
    b.H = A.s_height[i];
    b.P = A.s_players[i];
    bool bad = b.W < 1 || b.W > A.max_w || b.H < 1 || b.H > A.max_h || b.P < 1 || b.P > A.max_p || b.P > MAXP;
    if (bad) {
      if (lane == 0) atomicExch(A.status, GVEC_E_INVALID);
      return;
    }
    b.N = b.W * b.H;
    b.recipW = (65536 + b.W - 1) / b.W;
    b.turn = 0;
    b.hflags = A.fog ? HF_FOG : 0u;
    b.alive = (1u << b.P) - 1u;  // initializePlayers: Alive = true (engine_initializer.go:125-143)
    b.hv = (lane >= H_GIDX && lane < H_GIDX + 8) ? 0xFFFFFFFFu : 0u;  // GeneralIdx -1, counters / episode 0
#pragma unroll
    for (int p = 0; p < MAXP; ++p) b.own[p] = b.lst[p] = b.vis[p] = 0u;
    b.chg = b.vch = b.gen = b.city = b.mtn = 0u;
#pragma unroll
    for (int s = 0; s < NSLOT; ++s) b.army[s] = 0;
  } else {
    load_board(b, hdr, rows, army, A.fd);
  }
  b.geometry();

  // per-tile source planes are read coalesced in the tile domain (lane l, slot s = tile 64s+l);
  // each predicate becomes a flat plane through the wave ballot
  bool bad_owner = false;
  if (A.s_owner) {
#pragma unroll
    for (int p = 0; p < MAXP; ++p) b.own[p] = 0u;
#pragma unroll
    for (int s = 0; s < NSLOT; ++s) {
      const int t = 64 * s + lane;
      const int o = (t < b.N) ? (int)A.s_owner[to + t] : -1;
      bad_owner |= (o < -1) || (o >= b.P);
#pragma unroll
      for (int p = 0; p < MAXP; ++p) b.scatter(b.own[p], __builtin_amdgcn_ballot_w64(o == p), s);
    }
  }
  if (wave_any(bad_owner)) {
    if (lane == 0) atomicExch(A.status, GVEC_E_BOARD);
    return;
  }
  if (A.s_type) {
    b.gen = b.city = b.mtn = 0u;
#pragma unroll
    for (int s = 0; s < NSLOT; ++s) {
      const int t = 64 * s + lane;
      const int ty = (t < b.N) ? (int)A.s_type[to + t] : GVEC_TILE_NORMAL;
      b.scatter(b.gen, __builtin_amdgcn_ballot_w64(ty == GVEC_TILE_GENERAL), s);
      b.scatter(b.city, __builtin_amdgcn_ballot_w64(ty == GVEC_TILE_CITY), s);
      b.scatter(b.mtn, __builtin_amdgcn_ballot_w64(ty == GVEC_TILE_MOUNTAIN), s);
    }
  }
  if (A.s_visible) {
#pragma unroll
    for (int p = 0; p < MAXP; ++p) b.vis[p] = 0u;
#pragma unroll
    for (int s = 0; s < NSLOT; ++s) {
      const int t = 64 * s + lane;
      const uint32_t v = (t < b.N) ? (uint32_t)A.s_visible[to + t] : 0u;
#pragma unroll
      for (int p = 0; p < MAXP; ++p) b.scatter(b.vis[p], __builtin_amdgcn_ballot_w64(((v >> p) & 1u) != 0u), s);
    }
  }
  if (A.s_listed) {
#pragma unroll
    for (int p = 0; p < MAXP; ++p) b.lst[p] = 0u;
#pragma unroll
    for (int s = 0; s < NSLOT; ++s) {
      const int t = 64 * s + lane;
      const int o = (t < b.N) ? (int)A.s_listed[to + t] : -1;
#pragma unroll
      for (int p = 0; p < MAXP; ++p) b.scatter(b.lst[p], __builtin_amdgcn_ballot_w64(o == p), s);
    }
  }
  if (A.s_changed) {
    b.chg = 0u;
#pragma unroll
    for (int s = 0; s < NSLOT; ++s) {
      const int t = 64 * s + lane;
      b.scatter(b.chg, __builtin_amdgcn_ballot_w64(t < b.N && A.s_changed[to + t] != 0), s);
    }
  }
  if (A.s_vis_changed) {
    b.vch = 0u;
#pragma unroll
    for (int s = 0; s < NSLOT; ++s) {
      const int t = 64 * s + lane;
      b.scatter(b.vch, __builtin_amdgcn_ballot_w64(t < b.N && A.s_vis_changed[to + t] != 0), s);
    }
  }
  if (A.s_army) {
#pragma unroll
    for (int s = 0; s < NSLOT; ++s) {
      const int t = 64 * s + lane;
      b.army[s] = (t < b.N) ? A.s_army[to + t] : 0;
    }
  }
  if (A.s_turn) b.turn = A.s_turn[i];
  if (A.s_done) b.hflags = A.s_done[i] ? (b.hflags | HF_DONE) : (b.hflags & ~HF_DONE);
  if (A.s_alive) {
    uint32_t al = 0u;
    for (int p = 0; p < b.P; ++p) al |= (A.s_alive[po + p] ? 1u : 0u) << p;
    b.alive = al;
  }
  for (int p = 0; p < b.P; ++p) {
    if (A.s_army_count) b.hdr_set(H_ARMYCNT + p, (uint32_t)A.s_army_count[po + p]);
    if (A.s_general_idx) b.hdr_set(H_GIDX + p, (uint32_t)A.s_general_idx[po + p]);
  }
  // the planes that are functions of the board: rebuilt on every import (the type planes may have changed)
  b.targets();
  b.static_flags();
  b.refresh_gt1();
  if (A.init) b.hflags |= HF_SETUP;  // performInitialSetup runs in setup_kernel, on the turn engine's layout
  b.store_army(army);
  b.settle_lists();
  b.store_hdr(hdr, A.fresh ? 0u : ((b.hdr_get(H_STATUS) >> 16) & 0xFFu));
  b.store_planes(rows, A.fd, A.row_dw, true);
}

// =========================================================================================
// export: resident record -> planes (gvec_read_state / gvec_player_visibility)
// =========================================================================================
template <int MAXP, int NSLOT>
__global__ __launch_bounds__(64 * WAVES_PER_BLOCK) void export_kernel(ExportArgs A) {
  using B = Board<MAXP, NSLOT>;
  const int wave = (int)(threadIdx.x >> 6), lane = lane_id();
  const int i = uni((int)blockIdx.x * WAVES_PER_BLOCK + wave);
  if (i >= A.n) return;
  const int env = A.env_begin + i;
  B b;
  load_board(b, A.hdr + (size_t)env * HDR_DW, A.rows + (size_t)env * A.row_dw, army_cref<NSLOT>(A.army16, A.army32, env), A.fd);
  const size_t to = (size_t)i * A.stride, po = (size_t)i * A.max_p;
  const uint32_t special = b.gen | b.city | b.mtn;
  uint32_t pv_plane = 0u;
#pragma unroll
  for (int p = 0; p < MAXP; ++p) pv_plane = (p == A.vis_player) ? b.vis[p] : pv_plane;
  const bool fog_on = (b.hflags & HF_FOG) != 0u;
#pragma unroll
  for (int s = 0; s < NSLOT; ++s) {
    const int t = 64 * s + lane;
    const bool in = t < b.N;
    int owner = -1, listed = -1;
    uint32_t visb = 0u;
#pragma unroll
    for (int p = 0; p < MAXP; ++p) {
      owner = b.gather(b.own[p], s) ? p : owner;
      listed = b.gather(b.lst[p], s) ? p : listed;
      visb |= b.gather(b.vis[p], s) << p;
    }
    // every gather is a cross-lane ds_bpermute: evaluate them all convergently, never inside a
    // per-lane short-circuit (a masked-off source lane reads back as 0)
    const uint32_t is_gen = b.gather(b.gen, s), is_city = b.gather(b.city, s), is_mtn = b.gather(b.mtn, s);
    const int type = is_gen ? GVEC_TILE_GENERAL : (is_city ? GVEC_TILE_CITY : (is_mtn ? GVEC_TILE_MOUNTAIN : GVEC_TILE_NORMAL));
    const uint32_t c = b.gather(b.chg, s), vc = b.gather(b.vch, s);
    const uint32_t pv = b.gather(pv_plane, s), sp = b.gather(special, s);
    if (t < A.stride) {
      if (A.army_out) A.army_out[to + t] = in ? b.army[s] : 0;
      if (A.owner) A.owner[to + t] = (int8_t)(in ? owner : -1);
      if (A.type) A.type[to + t] = (uint8_t)(in ? type : 0);
      if (A.visible) A.visible[to + t] = (uint8_t)(in ? visb : 0u);
      if (A.listed) A.listed[to + t] = (int8_t)(in ? listed : -1);
      if (A.changed) A.changed[to + t] = (uint8_t)(in ? c : 0u);
      if (A.vis_changed) A.vis_changed[to + t] = (uint8_t)(in ? vc : 0u);
      // ComputePlayerVisibilityOptimized (visibility_optimized.go:166-195)
      if (A.pv_visible) A.pv_visible[to + t] = (uint8_t)(in ? (fog_on ? pv : 1u) : 0u);
      if (A.pv_fog) A.pv_fog[to + t] = (uint8_t)((in && fog_on && !pv && sp) ? 1u : 0u);
    }
  }
  uint32_t tcnt[MAXP];
#pragma unroll
  for (int p = 0; p < MAXP; ++p) tcnt[p] = wave_sum((uint32_t)__builtin_popcount(b.lst[p]));
  if (lane == 0) {
    if (A.turn) A.turn[i] = b.turn;
    if (A.done) A.done[i] = (uint8_t)((b.hflags & HF_DONE) ? 1 : 0);
    // Engine.GetWinner re-derives the winner from the CURRENT Alive flags (engine.go:248-263)
    const int na = __builtin_popcount(b.alive);
    if (A.winner) A.winner[i] = (int8_t)(((b.hflags & HF_DONE) && b.P > 1 && na == 1) ? (31 - __builtin_clz(b.alive)) : -1);
    if (A.width) A.width[i] = b.W;
    if (A.height) A.height[i] = b.H;
    if (A.players) A.players[i] = b.P;
  }
#pragma unroll
  for (int p = 0; p < MAXP; ++p) {
    if (lane == 0 && p < A.max_p) {
      const bool live = p < b.P;
      if (A.alive) A.alive[po + p] = (uint8_t)(live ? ((b.alive >> p) & 1u) : 0u);
      if (A.army_count) A.army_count[po + p] = live ? (int32_t)b.hdr_get(H_ARMYCNT + p) : 0;
      if (A.tile_count) A.tile_count[po + p] = live ? (int32_t)tcnt[p] : 0;
      if (A.general_idx) A.general_idx[po + p] = live ? (int32_t)b.hdr_get(H_GIDX + p) : -1;
    }
  }
}

// =========================================================================================
// resident records <-> canonical record slabs (gvec_export_records / gvec_import_records): a slab is
// [n][HDR_DW] headers | [n][row_dw] planes | [n][NSLOT*64] int32 armies - always the wide form, whatever the
// env's storage.  Import validates the header of every record before anything is trusted (a slab may come
// from another rank or from a file).
// =========================================================================================
template <int MAXP, int NSLOT, bool IMPORT>
__global__ __launch_bounds__(64 * WAVES_PER_BLOCK) void records_kernel(RecordArgs A) {
  using B = Board<MAXP, NSLOT>;
  const int wave = (int)(threadIdx.x >> 6), lane = lane_id();
  const int i = uni((int)blockIdx.x * WAVES_PER_BLOCK + wave);
  if (i >= A.n) return;
  const int env = A.env_begin + i;
  uint32_t* rec_hdr = A.rec_hdr + (size_t)i * HDR_DW;
  uint32_t* rec_rows = A.rec_rows + (size_t)i * A.row_dw;
  int32_t* rec_army = A.rec_army + (size_t)i * NSLOT * 64;
  B b;
  if constexpr (!IMPORT) {
    load_board(b, A.hdr + (size_t)env * HDR_DW, A.rows + (size_t)env * A.row_dw, army_cref<NSLOT>(A.army16, A.army32, env), A.fd);
    b.hflags &= ~HF_WIDE;
    army_store_wide<NSLOT>(b.army, rec_army);
    b.settle_lists();
    b.store_hdr(rec_hdr, (b.hdr_get(H_STATUS) >> 16) & 0xFFu);
    b.store_planes(rec_rows, A.fd, A.row_dw, true, true);  // a record carries its list planes whatever the flag says
  } else {
    b.load_hdr(rec_hdr);
    const bool bad = b.W < 1 || b.W > A.max_w || b.H < 1 || b.H > A.max_h || b.P < 1 || b.P > A.max_p || b.P > MAXP ||
                     b.recipW != (65536 + (b.W > 0 ? b.W : 1) - 1) / (b.W > 0 ? b.W : 1) || (b.alive >> b.P) != 0u;
    if (bad) {
      if (lane == 0) atomicExch(A.status, GVEC_E_BOARD);
      return;
    }
    b.hflags &= (HF_DONE | HF_FOG | HF_LDIFF);  // HF_LDIFF: where load_planes takes the lists from
    army_load_wide<NSLOT>(b.army, rec_army);
    b.load_planes(rec_rows, A.fd);
    b.geometry();  // the constant planes are rebuilt, never taken from the slab
    // nothing outside the board may be set: the turn logic relies on it
#pragma unroll
    for (int p = 0; p < MAXP; ++p) {
      const uint32_t keep = (p < b.P) ? b.valid : 0u;
      b.own[p] &= keep;
      b.lst[p] &= keep;
      b.vis[p] &= keep;
    }
    b.chg &= b.valid;
    b.vch &= b.valid;
    b.gen &= b.valid;
    b.city &= b.valid;
    b.mtn &= b.valid;
#pragma unroll
    for (int s = 0; s < NSLOT; ++s) b.army[s] = (64 * s + lane < b.N) ? b.army[s] : 0;
    b.targets();
    b.static_flags();
    b.refresh_gt1();
    b.store_army(army_ref<NSLOT>(A.army16, A.army32, env));
    b.settle_lists();
    b.store_hdr(A.hdr + (size_t)env * HDR_DW, (b.hdr_get(H_STATUS) >> 16) & 0xFFu);
    b.store_planes(A.rows + (size_t)env * A.row_dw, A.fd, A.row_dw, true);
  }
}

// =========================================================================================
// map generator: algorithm and ratios of mapgen/generator.go:25-253 on the counter RNG.
// One thread per board (reset-time work, sequential by nature); mirrored by ora_mapgen.
// =========================================================================================
struct MRng {
  uint32_t key, ctr;
  __device__ uint32_t draw() { return fmix32(key + (ctr++) * 0x9E3779B9u); }
  __device__ int intn(int n) { return (int)__umulhi(draw(), (uint32_t)n); }
  __device__ int shuf(int n) { return intn(n); }
  __device__ void begin(const MapgenArgs& A, int i) {
    key = fmix32(env_key(A.seed_lo, A.seed_hi, (uint32_t)(A.first_index + i)) ^ 0x5BD1E995u);
    ctr = 0u;
  }
};

// Go's math/rand - rand.New(rand.NewSource(seed)), go 1.24 - one generator per thread: the additive lagged Fibonacci
// generator x[n] = x[n-607] + x[n-273] mod 2^64 seeded by the LCG x = 48271 x mod (2^31 - 1) XOR the 607-word table
// (derived by scripts/gen_go_rand_cooked.py, not copied).  The 607-word state lives in a caller-provided global buffer,
// word k of thread i at vec[k * stride] (threads seed in lock-step: coalesced).  Mirrored by the oracle's ora_gorand,
// which the reference's own seed-12345 vectors pin (tests/test_go_rand.py).
__device__ const uint64_t go_rng_cooked[607] = {
#include "go_rand_cooked.inc"
};
struct GoRng {
  uint64_t* vec;
  size_t stride;
  int tap, feed;
  static __device__ int32_t seedrand(int32_t x) {
    const int32_t hi = x / 44488, lo = x % 44488;
    x = 48271 * lo - 3399 * hi;
    return x < 0 ? x + 2147483647 : x;
  }
  __device__ void seed(int64_t s) {
    tap = 0;
    feed = 607 - 273;
    s %= 2147483647ll;
    if (s < 0) s += 2147483647ll;
    if (s == 0) s = 89482311ll;
    int32_t x = (int32_t)s;
    for (int i = -20; i < 607; ++i) {
      x = seedrand(x);
      if (i >= 0) {
        uint64_t u = (uint64_t)x << 40;
        x = seedrand(x);
        u ^= (uint64_t)x << 20;
        x = seedrand(x);
        u ^= (uint64_t)x;
        vec[(size_t)i * stride] = u ^ go_rng_cooked[i];
      }
    }
  }
  __device__ uint64_t int63() {
    if (--tap < 0) tap += 607;
    if (--feed < 0) feed += 607;
    const uint64_t x = vec[(size_t)feed * stride] + vec[(size_t)tap * stride];
    vec[(size_t)feed * stride] = x;
    return x & 0x7FFFFFFFFFFFFFFFull;
  }
  __device__ int intn(int n) {  // Intn -> Int31n
    if ((n & (n - 1)) == 0) return (int)(int63() >> 32) & (n - 1);
    const int32_t mx = (int32_t)(2147483647u - (2147483648u % (uint32_t)n));
    int32_t v = (int32_t)(int63() >> 32);
    while (v > mx) v = (int32_t)(int63() >> 32);
    return v % n;
  }
  __device__ int shuf(int n) {  // rand.go int31n (Shuffle): Lemire's multiply-shift on Uint32
    uint32_t v = (uint32_t)(int63() >> 31);
    uint64_t prod = (uint64_t)v * (uint64_t)(uint32_t)n;
    uint32_t low = (uint32_t)prod;
    if (low < (uint32_t)n) {
      const uint32_t thresh = (uint32_t)(-n) % (uint32_t)n;
      while (low < thresh) {
        v = (uint32_t)(int63() >> 31);
        prod = (uint64_t)v * (uint64_t)(uint32_t)n;
        low = (uint32_t)prod;
      }
    }
    return (int)(prod >> 32);
  }
  __device__ void begin(const MapgenArgs& A, int i) {
    vec = A.go_state + i;
    stride = (size_t)A.n;
    seed(A.go_seeds[i]);
  }
};

template <typename RNG>
__device__ __forceinline__ void mapgen_board(RNG& r, const MapgenArgs& A, int i) {
  const int w = A.in_width ? A.in_width[i] : A.max_w, h = A.in_height ? A.in_height[i] : A.max_h;
  const int players = A.in_players ? A.in_players[i] : A.max_p;
  A.width[i] = w;
  A.height[i] = h;
  A.players[i] = players;
  if (w < 1 || w > A.max_w || h < 1 || h > A.max_h || players < 1 || players > A.max_p) {
    atomicExch(A.status, GVEC_E_INVALID);
    return;
  }
  int32_t* army = A.army + (size_t)i * A.stride;
  int8_t* owner = A.owner + (size_t)i * A.stride;
  uint8_t* type = A.type + (size_t)i * A.stride;
  const int n = w * h;
  r.begin(A, i);
  for (int t = 0; t < A.stride; ++t) {
    army[t] = 0;
    owner[t] = -1;
    type[t] = GVEC_TILE_NORMAL;
  }
  // DefaultMapConfig (generator.go:25-47; config.go:198-200)
  int spacing = 5;
  if (spacing > w / 2 + h / 2) spacing = w / 2 + h / 2;
  const int veins = n / 50, min_len = 3, max_len = w / 4, city_ratio = 20, city_army = 40;
  for (int v = 0; v < veins; ++v) {  // placeMountains :77-142
    int cx = -1, cy = -1;
    for (int a = 0; a < 100; ++a) {
      const int x = r.intn(w), y = r.intn(h);
      const int idx = y * w + x;
      if (type[idx] == GVEC_TILE_NORMAL && owner[idx] == -1) {
        cx = x;
        cy = y;
        break;
      }
    }
    if (cx < 0) continue;
    type[cy * w + cx] = GVEC_TILE_MOUNTAIN;
    int len = min_len;
    if (max_len > min_len) len += r.intn(max_len - min_len + 1);
    for (int k = 1; k < len; ++k) {
      // dirs packed 2 bits each, N E S W = 0 1 2 3; rand.Shuffle = Fisher-Yates from the top (:117)
      uint32_t dirs = 0xE4u;  // [0]=0,[1]=1,[2]=2,[3]=3
      for (int a = 3; a > 0; --a) {
        const int j = r.shuf(a + 1);
        const uint32_t da = (dirs >> (2 * a)) & 3u, dj = (dirs >> (2 * j)) & 3u;
        dirs = (dirs & ~((3u << (2 * a)) | (3u << (2 * j))));
        dirs |= (dj << (2 * a)) | (da << (2 * j));
      }
      uint64_t cand = 0ull;  // candidate (x,y) pairs packed 10 bits each, in shuffled-direction order
      int nc = 0;
      for (int j = 0; j < 4; ++j) {
        const int d = (int)((dirs >> (2 * j)) & 3u);
        const int nx = cx + ((d == 1) - (d == 3)), ny = cy + ((d == 2) - (d == 0));
        if (nx >= 0 && nx < w && ny >= 0 && ny < h) {
          const int ni = ny * w + nx;
          if (type[ni] == GVEC_TILE_NORMAL && owner[ni] == -1) {
            cand |= (uint64_t)(uint32_t)(nx | (ny << 5)) << (10 * nc);
            nc++;
          }
        }
      }
      if (nc == 0) break;
      const int pick = r.intn(nc);
      const uint32_t c = (uint32_t)(cand >> (10 * pick)) & 1023u;
      cx = (int)(c & 31u);
      cy = (int)(c >> 5);
      type[cy * w + cx] = GVEC_TILE_MOUNTAIN;
    }
  }
  {  // placeCities :144-164
    const int want = n / city_ratio, max_attempts = want * 20;
    int placed = 0, attempts = 0;
    while (placed < want && attempts < max_attempts) {
      const int x = r.intn(w), y = r.intn(h);
      const int idx = y * w + x;
      if (owner[idx] == -1 && type[idx] == GVEC_TILE_NORMAL) {
        type[idx] = GVEC_TILE_CITY;
        army[idx] = city_army;
        placed++;
      }
      attempts++;
    }
  }
  int gx[GVEC_MAX_PLAYERS], gy[GVEC_MAX_PLAYERS];
  for (int pid = 0; pid < players; ++pid) {  // placeGenerals :166-253
    int placed_idx = -1;
    for (int a = 0; a < n && placed_idx < 0; ++a) {
      const int x = r.intn(w), y = r.intn(h);
      const int idx = y * w + x;
      if (owner[idx] != -1 || type[idx] != GVEC_TILE_NORMAL) continue;
      bool ok = true;
      for (int o = 0; o < pid; ++o) ok = ok && (abs(x - gx[o]) + abs(y - gy[o]) >= spacing);
      if (ok) placed_idx = idx;
    }
    for (int idx = 0; idx < n && placed_idx < 0; ++idx) {  // fallback scan :223-250
      if (owner[idx] != -1 || type[idx] != GVEC_TILE_NORMAL) continue;
      const int x = idx % w, y = idx / w;
      bool ok = true;
      for (int o = 0; o < pid; ++o) ok = ok && (abs(x - gx[o]) + abs(y - gy[o]) >= spacing);
      if (ok) placed_idx = idx;
    }
    if (placed_idx < 0) {
      atomicExch(A.status, GVEC_E_BOARD);
      return;
    }
    owner[placed_idx] = (int8_t)pid;
    army[placed_idx] = 2;
    type[placed_idx] = GVEC_TILE_GENERAL;
    gx[pid] = placed_idx % w;
    gy[pid] = placed_idx / w;
  }
}

__global__ void mapgen_kernel(MapgenArgs A) {
  const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (i >= A.n) return;
  MRng r;
  mapgen_board(r, A, i);
}
// the same generator on Go's math/rand: board i = what game.NewEngine builds from GameConfig.Rng = rand.New(rand.NewSource(go_seeds[i]))
__global__ void mapgen_go_kernel(MapgenArgs A) {
  const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (i >= A.n) return;
  GoRng r;
  mapgen_board(r, A, i);
}

// =========================================================================================
__global__ void counter_sum_kernel(const uint32_t* hdr, int32_t num_envs, unsigned long long* out) {
  unsigned long long s0 = 0, s1 = 0, s2 = 0;
  for (int e = (int)(blockIdx.x * blockDim.x + threadIdx.x); e < num_envs; e += (int)(gridDim.x * blockDim.x)) {
    const uint32_t* h = hdr + (size_t)e * HDR_DW;
    s0 += h[H_CNT_STEPS];
    s1 += h[H_CNT_ABORT];
    s2 += h[H_CNT_DONE];
  }
  for (int off = 32; off > 0; off >>= 1) {
    s0 += __shfl_down(s0, off);
    s1 += __shfl_down(s1, off);
    s2 += __shfl_down(s2, off);
  }
  if ((threadIdx.x & 63u) == 0u) {
    atomicAdd(out + 0, s0);
    atomicAdd(out + 1, s1);
    atomicAdd(out + 2, s2);
  }
}

// device self-test of the wave primitives the engine relies on (run by tests / smoke)
__global__ void selftest_kernel(int32_t* out) {
  const int lane = lane_id();
  int fail = 0;
  const uint32_t v = (uint32_t)(lane * 3 + 1);
  if (from_prev(v) != (lane == 0 ? 0u : (uint32_t)((lane - 1) * 3 + 1))) fail = 1;
  if (from_next(v) != (lane == 63 ? 0u : (uint32_t)((lane + 1) * 3 + 1))) fail = fail ? fail : 2;
  uint32_t expect = 0u;
  for (int l = 0; l <= lane; ++l) expect += (uint32_t)(l * 3 + 1);
  if (wave_scan_add(v) != expect) fail = fail ? fail : 3;
  if (wave_sum(v) != (uint32_t)(63 * 64 / 2 * 3 + 64)) fail = fail ? fail : 4;
  if (bperm(4 * ((lane * 7) & 63), v) != (uint32_t)(((lane * 7) & 63) * 3 + 1)) fail = fail ? fail : 5;
  {  // per-lane k-th set bit: lane l asks for bit number l % popcount
    const uint32_t w = 0x80000105u ^ ((uint32_t)lane * 0x9E3779B1u);
    const uint32_t r = (uint32_t)lane % (uint32_t)__builtin_popcount(w);
    uint32_t want = 0u, seen = 0u;
    for (uint32_t i = 0; i < 32u; ++i)
      if ((w >> i) & 1u) {
        if (seen == r) want = i;
        ++seen;
      }
    if (kth_set_bit(w, r) != want) fail = fail ? fail : 7;
  }
  {  // row-wise scans and the row-last broadcast, 16- and 32-lane rows
    uint32_t e16 = 0u, e32 = 0u, o16 = 0u, o32 = 0u;
    for (int l = lane & ~15; l <= lane; ++l) e16 += (uint32_t)(l * 3 + 1), o16 |= 1u << (l & 31);
    for (int l = lane & ~31; l <= lane; ++l) e32 += (uint32_t)(l * 3 + 1), o32 |= 1u << (l & 31);
    if (row_scan_add<16>(v) != e16 || row_scan_add<32>(v) != e32) fail = fail ? fail : 8;
    if (row_scan_or<16>(1u << (lane & 31)) != o16 || row_scan_or<32>(1u << (lane & 31)) != o32) fail = fail ? fail : 9;
    if (row_last<16>(v) != (uint32_t)((lane | 15) * 3 + 1) || row_last<32>(v) != (uint32_t)((lane | 31) * 3 + 1)) fail = fail ? fail : 10;
  }
  if ((uint32_t)gvec_llvm_writelane(777, 5, (int)v) != (lane == 5 ? 777u : v)) fail = fail ? fail : 11;
  if (mad24(v, 3u, 5u) != v * 3u + 5u) fail = fail ? fail : 12;
  const unsigned long long any = __builtin_amdgcn_ballot_w64(fail != 0);
  if (lane == 0) out[0] = any ? (int32_t)(__builtin_ctzll(any) * 16 + rdlane((uint32_t)fail, (int)__builtin_ctzll(any))) : 0;
}

// =========================================================================================
// host-side dispatch
// =========================================================================================
bool pick_variant(int max_players, int tile_stride, Variant* out) {
  static const int kP[] = {2, 4, 8};
  static const int kS[] = {1, 2, 4, 7, 10, 16};
  int need = (tile_stride + 63) / 64;
  out->maxp = 0;
  out->nslot = 0;
  for (int p : kP)
    if (max_players <= p) {
      out->maxp = p;
      break;
    }
  for (int s : kS)
    if (need <= s) {
      out->nslot = s;
      break;
    }
  return out->maxp && out->nslot;
}

template <typename F>
static hipError_t dispatch(const Variant& v, F&& f) {
#define GVEC_CASE(P_, S_) \
  if (v.maxp == P_ && v.nslot == S_) return f(std::integral_constant<int, P_>{}, std::integral_constant<int, S_>{});
#define GVEC_ROW(P_) GVEC_CASE(P_, 1) GVEC_CASE(P_, 2) GVEC_CASE(P_, 4) GVEC_CASE(P_, 7) GVEC_CASE(P_, 10) GVEC_CASE(P_, 16)
  GVEC_ROW(2) GVEC_ROW(4) GVEC_ROW(8)
#undef GVEC_ROW
#undef GVEC_CASE
  return hipErrorInvalidValue;
}

static inline dim3 wave_grid(int n) { return dim3((unsigned)((n + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK)); }

// env_key_of(base, env) = fmix32(base + env * C): a handle that is shard [env_base, env_base + B) of a larger batch
// (gvec_create_sharded) folds its offset into the bases, and its env e then draws exactly what env env_base + e of one
// big handle would - agent moves and pool boards alike; the kernels never see the offset.
static inline StepArgs with_seed_bases(const StepArgs& in) {
  StepArgs a = in;
  a.seed_base = env_key_base(a.seed_lo, a.seed_hi) + (uint32_t)a.env_base * 0xC2B2AE3Du;
  a.pool_seed_base = env_key_base(a.pool_seed_lo, a.pool_seed_hi) + (uint32_t)a.env_base * 0xC2B2AE3Du;
  return a;
}

hipError_t launch_step(const Variant& v, const StepArgs& in, hipStream_t s) {
  const StepArgs a = with_seed_bases(in);
  return dispatch(v, [&](auto P_, auto S_) {
    constexpr int P = decltype(P_)::value, S = decltype(S_)::value;
    // the resident format keeps planes of 2*S-1 or 2*S dwords (gvec_api.hip: plane_dwords)
    const bool odd = a.fd == 2 * S - 1, agent = (a.flags & KF_AGENT) != 0u;
    if ((!odd && a.fd != 2 * S) || a.row_dw != (Planes<P>::COUNT * a.fd + 3) / 4 * 4) return hipErrorInvalidValue;
    const dim3 grid = wave_grid(a.num_envs), block(64 * WAVES_PER_BLOCK);
    if (agent && odd) hipLaunchKernelGGL((step_kernel<P, S, true, true>), grid, block, 0, s, a);
    else if (agent) hipLaunchKernelGGL((step_kernel<P, S, true, false>), grid, block, 0, s, a);
    else if (odd) hipLaunchKernelGGL((step_kernel<P, S, false, true>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((step_kernel<P, S, false, false>), grid, block, 0, s, a);
    return hipGetLastError();
  });
}
hipError_t launch_rollout(const Variant& v, const StepArgs& in, hipStream_t s) {
  const StepArgs a = with_seed_bases(in);
  return dispatch(v, [&](auto P_, auto S_) {
    constexpr int P = decltype(P_)::value, S = decltype(S_)::value;
    hipLaunchKernelGGL((rollout_kernel<P, S>), wave_grid(a.num_envs), dim3(64 * WAVES_PER_BLOCK), 0, s, a);
    return hipGetLastError();
  });
}
hipError_t launch_agent(const Variant& v, const StepArgs& in, hipStream_t s) {
  const StepArgs a = with_seed_bases(in);
  return dispatch(v, [&](auto P_, auto S_) {
    hipLaunchKernelGGL((query_kernel<decltype(P_)::value, decltype(S_)::value, 1>), wave_grid(a.num_envs),
                       dim3(64 * WAVES_PER_BLOCK), 0, s, a);
    return hipGetLastError();
  });
}
hipError_t launch_legal(const Variant& v, const StepArgs& a, hipStream_t s) {
  return dispatch(v, [&](auto P_, auto S_) {
    hipLaunchKernelGGL((query_kernel<decltype(P_)::value, decltype(S_)::value, 0>), wave_grid(a.num_envs),
                       dim3(64 * WAVES_PER_BLOCK), 0, s, a);
    return hipGetLastError();
  });
}
hipError_t launch_serializer_mask(const Variant& v, const StepArgs& a, hipStream_t s) {
  return dispatch(v, [&](auto P_, auto S_) {
    hipLaunchKernelGGL((query_kernel<decltype(P_)::value, decltype(S_)::value, 2>), wave_grid(a.num_envs), dim3(64 * WAVES_PER_BLOCK),
                       0, s, a);
    return hipGetLastError();
  });
}
hipError_t launch_snapshot(const Variant& v, const ExperienceArgs& a, hipStream_t s) {
  return dispatch(v, [&](auto P_, auto S_) {
    hipLaunchKernelGGL((snapshot_kernel<decltype(P_)::value, decltype(S_)::value>), wave_grid(a.num_envs), dim3(64 * WAVES_PER_BLOCK),
                       0, s, a);
    return hipGetLastError();
  });
}
hipError_t launch_rewards(const Variant& v, const ExperienceArgs& a, hipStream_t s) {
  return dispatch(v, [&](auto P_, auto S_) {
    hipLaunchKernelGGL((rewards_kernel<decltype(P_)::value, decltype(S_)::value>), wave_grid(a.num_envs), dim3(64 * WAVES_PER_BLOCK),
                       0, s, a);
    return hipGetLastError();
  });
}
hipError_t launch_experience_records(const Variant& v, const ExperienceArgs& a, hipStream_t s) {
  return dispatch(v, [&](auto P_, auto S_) {
    hipLaunchKernelGGL((experience_record_kernel<decltype(P_)::value, decltype(S_)::value>), wave_grid(a.num_envs),
                       dim3(64 * WAVES_PER_BLOCK), 0, s, a);
    return hipGetLastError();
  });
}
void experience_layout(const Variant& v, int fd, int* snap_dw, int* record_dw) {
  (void)dispatch(v, [&](auto P_, auto S_) {
    constexpr int P = decltype(P_)::value, S = decltype(S_)::value;
    *snap_dw = SnapLayout<P, S>{fd}.total();
    *record_dw = RecordLayout<P, S>{fd}.total();
    return hipSuccess;
  });
}
hipError_t launch_gym_observe(const Variant& v, const GymArgs& a, hipStream_t s) {
  return dispatch(v, [&](auto P_, auto S_) {
    hipLaunchKernelGGL((gym_observe_kernel<decltype(P_)::value, decltype(S_)::value>), wave_grid(a.num_envs), dim3(64 * WAVES_PER_BLOCK),
                       0, s, a);
    return hipGetLastError();
  });
}
hipError_t launch_gym_step(const Variant& v, const StepArgs& in, const GymStepArgs& g, hipStream_t s) {
  const StepArgs a = with_seed_bases(in);
  return dispatch(v, [&](auto P_, auto S_) {
    constexpr int P = decltype(P_)::value, S = decltype(S_)::value;
    const bool odd = a.fd == 2 * S - 1;
    if ((!odd && a.fd != 2 * S) || a.row_dw != (Planes<P>::COUNT * a.fd + 3) / 4 * 4) return hipErrorInvalidValue;
    const dim3 grid = wave_grid(a.num_envs), block(64 * WAVES_PER_BLOCK);
    if (odd) hipLaunchKernelGGL((gym_step_kernel<P, S, true>), grid, block, 0, s, a, g);
    else hipLaunchKernelGGL((gym_step_kernel<P, S, false>), grid, block, 0, s, a, g);
    return hipGetLastError();
  });
}
hipError_t launch_stream_deltas(const Variant& v, const StreamDeltaArgs& a, hipStream_t s) {
  return dispatch(v, [&](auto P_, auto S_) {
    hipLaunchKernelGGL((stream_delta_kernel<decltype(P_)::value, decltype(S_)::value>), wave_grid(a.num_envs), dim3(64 * WAVES_PER_BLOCK), 0, s, a);
    return hipGetLastError();
  });
}
hipError_t launch_pack_updates(const unsigned long long* rows, const int32_t* count, long long* offset, unsigned long long* packed, int32_t n,
                               int32_t cap, long long capacity, hipStream_t s) {
  hipLaunchKernelGGL(scan_counts_kernel, dim3(1), dim3(1024), 0, s, count, offset, n);
  hipLaunchKernelGGL(pack_updates_kernel, wave_grid(n), dim3(64 * WAVES_PER_BLOCK), 0, s, rows, count, offset, packed, n, cap, capacity);
  return hipGetLastError();
}
size_t pool_collect_scratch_bytes(int32_t n) {
  const size_t g = (size_t)collect_groups(n);
  return (2 * (g + 1) + 2) * 8 + (size_t)n * 8 + (size_t)n * 4 + g * 64;
}
hipError_t launch_pool_collect(const gvec_collect_args& a, hipStream_t s) {
  const int padded = collect_groups(a.num_envs) * 64;
  hipLaunchKernelGGL(collect_flags_kernel, dim3((unsigned)((padded + 255) / 256)), dim3(256), 0, s, a);
  hipLaunchKernelGGL(collect_scan_kernel, dim3(1), dim3(1024), 0, s, a);
  int shift = 0;                                   // enough wavefronts to fill 256 CUs when there are few workers
  while (shift < 3 && ((long long)a.num_envs << shift) < 16384) ++shift;
  hipLaunchKernelGGL(collect_push_kernel, wave_grid(a.num_envs << shift), dim3(64 * WAVES_PER_BLOCK), 0, s, a, shift);
  return hipGetLastError();
}
hipError_t launch_gym_actions(const GymActArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(gym_actions_kernel, dim3((unsigned)((a.num_envs + 255) / 256)), dim3(256), 0, s, a);
  return hipGetLastError();
}
hipError_t launch_expand_records(const void* records, int32_t n, const int32_t* layout8, float* state, float* next_state, uint8_t* mask,
                                 int32_t* meta, hipStream_t s) {
  ExpandArgs a;
  a.records = reinterpret_cast<const uint32_t*>(records);
  a.state = state;
  a.next_state = next_state;
  a.mask = mask;
  a.meta = meta;
  a.n = n;
  a.record_dw = layout8[0];
  a.mp = layout8[1];
  a.fd = layout8[2];
  a.ns = layout8[3];
  a.stride = layout8[5];
  const size_t lds = (size_t)WAVES_PER_BLOCK * (a.record_dw + 2 * a.fd) * 4;   // <= 52 KB (32x32 8P)
  hipLaunchKernelGGL(expand_records_kernel, wave_grid(n), dim3(64 * WAVES_PER_BLOCK), lds, s, a);
  return hipGetLastError();
}
hipError_t launch_observe(const Variant& v, const ExperienceArgs& a, hipStream_t s) {
  return dispatch(v, [&](auto P_, auto S_) {
    hipLaunchKernelGGL((observe_kernel<decltype(P_)::value, decltype(S_)::value>), wave_grid(a.num_envs), dim3(64 * WAVES_PER_BLOCK),
                       0, s, a);
    return hipGetLastError();
  });
}
hipError_t launch_import(const Variant& v, const ImportArgs& a, hipStream_t s) {
  return dispatch(v, [&](auto P_, auto S_) {
    hipLaunchKernelGGL((import_kernel<decltype(P_)::value, decltype(S_)::value>), wave_grid(a.n), dim3(64 * WAVES_PER_BLOCK),
                       0, s, a);
    return hipGetLastError();
  });
}
hipError_t launch_setup(const Variant& v, const ImportArgs& a, hipStream_t s) {
  return dispatch(v, [&](auto P_, auto S_) {
    hipLaunchKernelGGL((setup_kernel<decltype(P_)::value, decltype(S_)::value>), wave_grid(a.n), dim3(64 * WAVES_PER_BLOCK),
                       0, s, a);
    return hipGetLastError();
  });
}
hipError_t launch_export(const Variant& v, const ExportArgs& a, hipStream_t s) {
  return dispatch(v, [&](auto P_, auto S_) {
    hipLaunchKernelGGL((export_kernel<decltype(P_)::value, decltype(S_)::value>), wave_grid(a.n), dim3(64 * WAVES_PER_BLOCK),
                       0, s, a);
    return hipGetLastError();
  });
}
hipError_t launch_records(const Variant& v, const RecordArgs& a, bool import, hipStream_t s) {
  return dispatch(v, [&](auto P_, auto S_) {
    constexpr int P = decltype(P_)::value, S = decltype(S_)::value;
    if (import) hipLaunchKernelGGL((records_kernel<P, S, true>), wave_grid(a.n), dim3(64 * WAVES_PER_BLOCK), 0, s, a);
    else hipLaunchKernelGGL((records_kernel<P, S, false>), wave_grid(a.n), dim3(64 * WAVES_PER_BLOCK), 0, s, a);
    return hipGetLastError();
  });
}
hipError_t launch_mapgen(const MapgenArgs& a, hipStream_t s) {
  if (a.go_seeds) hipLaunchKernelGGL(mapgen_go_kernel, dim3((unsigned)((a.n + 63) / 64)), dim3(64), 0, s, a);
  else hipLaunchKernelGGL(mapgen_kernel, dim3((unsigned)((a.n + 63) / 64)), dim3(64), 0, s, a);
  return hipGetLastError();
}
hipError_t launch_counter_sum(const uint32_t* hdr, int32_t num_envs, unsigned long long* out, hipStream_t s) {
  int blocks = (num_envs + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(counter_sum_kernel, dim3((unsigned)blocks), dim3(256), 0, s, hdr, num_envs, out);
  return hipGetLastError();
}
hipError_t launch_selftest(int32_t* out, hipStream_t s) {
  hipLaunchKernelGGL(selftest_kernel, dim3(1), dim3(64), 0, s, out);
  return hipGetLastError();
}

}  // namespace gvec
