/*
 * generals_vec.h — C ABI of the MI355X-native batched Generals.io turn engine.
 *
 * This is the drop-in boundary for ONE hot path of
 * mitchelldurbincs/GeneralsReinforcementLearning: the per-game turn loop of
 * internal/game + internal/game/core (movement/combat, production, 3x3 fog of
 * war, legal-action mask), batched over B independent boards ("envs").
 *
 * The reference has no FFI seam; the seam is the Go method set of *game.Engine
 * (internal/game/engine.go:62-298).  Every entry point below names the Go
 * symbol(s) it replaces.  A cgo shim binds these 1:1.
 *
 * Conventions
 *   - All functions return int32 status: 0 = GVEC_OK, negative = API misuse /
 *     runtime failure (never a game-rule error).  Per-env game-rule errors are
 *     delivered in int32 err[B] using the sentinel numbering of
 *     internal/game/core/errors.go:8-17 == proto/common/v1/common.proto:39-48.
 *   - Buffers are caller-owned, env-major.  `mem` says where they live:
 *     GVEC_MEM_HOST (the library copies through its own device staging buffers) or
 *     GVEC_MEM_DEVICE (pointers are HIP device pointers on the handle's device;
 *     no host copy, work is enqueued on the handle's stream and NOT synchronised).
 *   - Tile planes use the reference's row-major index  t = y*W + x
 *     (core/board.go:108) with the env's OWN width W, packed at the start of a
 *     slot of max_width*max_height elements (padded batch, SURVEY config 5).
 *   - A handle is not thread-safe (mirrors Engine: externally serialised,
 *     internal/grpc/gameserver/game_manager.go:576-602).
 *   - A handle lives on ONE device (gvec_create) or spans several (gvec_create_sharded).
 *   - There is NO CPU fallback behind this ABI: every compute entry point runs
 *     hand-written HIP kernels for gfx950 and fails with GVEC_E_NO_DEVICE when no
 *     GPU is present.
 */
#ifndef GENERALS_VEC_H
#define GENERALS_VEC_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GVEC_ABI_VERSION 2

/* limits of this build */
#define GVEC_MAX_PLAYERS 8   /* VisibleBitfield is carried as a u8 per tile            */
#define GVEC_MAX_DIM     32  /* one board row is one u32 bit-row on the device          */

/* API status (negative) */
#define GVEC_OK            0
#define GVEC_E_INVALID    -1  /* bad argument                                           */
#define GVEC_E_NO_DEVICE  -2  /* no HIP device / device init failed                     */
#define GVEC_E_HIP        -3  /* HIP runtime error (see gvec_last_error)                */
#define GVEC_E_RANGE      -4  /* env range / size out of bounds                         */
#define GVEC_E_BOARD      -5  /* reset/write_state input violates the board contract    */

/* per-env game errors (core/errors.go:8-17; common.proto:39-48) */
#define GVEC_ERR_NONE                 0
#define GVEC_ERR_INVALID_COORDINATES  1
#define GVEC_ERR_NOT_ADJACENT         2
#define GVEC_ERR_NOT_OWNED            3
#define GVEC_ERR_INSUFFICIENT_ARMY    4
#define GVEC_ERR_GAME_OVER            5
#define GVEC_ERR_INVALID_PLAYER       6
#define GVEC_ERR_MOVE_TO_SELF         7
#define GVEC_ERR_TARGET_IS_MOUNTAIN   8

/* tile types (core/board.go:20-26) */
#define GVEC_TILE_NORMAL   0
#define GVEC_TILE_GENERAL  1
#define GVEC_TILE_CITY     2
#define GVEC_TILE_MOUNTAIN 3
#define GVEC_NEUTRAL      (-1)

#define GVEC_MEM_HOST   0
#define GVEC_MEM_DEVICE 1

/* One player's action for one turn == *core.MoveAction (core/action.go:23-36)
 * after convertProtoAction (internal/grpc/gameserver/converters.go:105-132).
 * Coordinates are clamped Go ints: any out-of-board value stays out of board,
 * so MoveAction.Validate's ErrInvalidCoordinates branch is preserved.
 * The slot index in actions[env][player] is MoveAction.PlayerID. */
#define GVEC_ACT_VALID 1u   /* 0 = nil action (no-op this turn)                        */
#define GVEC_ACT_HALF  2u   /* proto Action.half; MoveAll = !half                      */
#define GVEC_ACT_SKIP_ENV 4u /* on actions[env][0] only: this env does not play a turn in
                               this call (err 0, state untouched).  Lets a vector env
                               mirror clients whose games advance at different times
                               (python/generals_gym/generals_env.py:226-241 returns
                               before submitting anything on an invalid action).         */
#define GVEC_ACT_RESET_ENV 8u /* on actions[env][0] only, handles with auto_reset and a board pool: this env
                                is re-dealt from the pool in this call whether or not its game is over
                                (err 0) - how a vector env ends a truncated episode
                                (generals_env.py:283-284: truncated = turn_count >= max_turns) without
                                reading `done` back and poking it.  Ignored without a pool.          */
typedef struct gvec_action {
  int8_t  from_x, from_y, to_x, to_y;
  uint8_t flags;
  uint8_t reserved[3];
} gvec_action; /* 8 bytes */

typedef struct gvec_config {
  int32_t abi_version;            /* GVEC_ABI_VERSION                                    */
  int32_t num_envs;               /* B                                                   */
  int32_t max_width, max_height;  /* <= GVEC_MAX_DIM each                                */
  int32_t max_players;            /* <= GVEC_MAX_PLAYERS                                 */
  int32_t device;                 /* HIP device ordinal                                  */
  int32_t fog_of_war;             /* GameState.FogOfWarEnabled (engine_initializer.go:118 = 1) */
  int32_t prod_general;           /* config.go:206  game.production.general = 1          */
  int32_t prod_city;              /* config.go:207  = 1                                  */
  int32_t prod_normal;            /* config.go:208  = 1                                  */
  int32_t normal_growth_interval; /* config.go:209  = 25                                 */
  int32_t auto_reset;             /* 1: a finished env is re-dealt from the board pool on
                                     its next step (vector-env semantics; no Go analogue) */
  int32_t reserved[4];
} gvec_config;

typedef struct gvec_handle gvec_handle;

/* Optional-plane view used by gvec_read_state / gvec_write_state.  NULL = skip.
 * Shapes: tile planes [n][max_width*max_height]; per-player [n][max_players];
 * scalars [n]. */
typedef struct gvec_state_view {
  int32_t* army;         /* Tile.Army (core/board.go:9).  A Go int is 64-bit; this engine computes in
                            int32 and stores exactly (u16 or int32 per env): a board whose armies
                            pass 2^31-1 would wrap here and not in Go - unreachable by play
                            (production adds 1 per tile per turn), reachable by gvec_write_state */
  int8_t*  owner;        /* Tile.Owner, -1 neutral         (core/board.go:8)            */
  uint8_t* type;         /* Tile.Type                      (core/board.go:10)           */
  uint8_t* visible;      /* Tile.VisibleBitfield, bit p    (core/board.go:11)           */
  int8_t*  listed;       /* p iff t in Players[p].OwnedTiles, else -1 (game/state.go:12)*/
  uint8_t* changed;      /* 1 iff t in GameState.ChangedTiles          (state.go:29)    */
  uint8_t* vis_changed;  /* 1 iff t in GameState.VisibilityChangedTiles (state.go:33)   */
  int32_t* turn;         /* GameState.Turn                                              */
  uint8_t* done;         /* Engine.gameOver                 (engine.go:20)              */
  int8_t*  winner;       /* Engine.GetWinner()              (engine.go:248-263)         */
  int32_t* width;        /* Board.W                                                     */
  int32_t* height;       /* Board.H                                                     */
  int32_t* players;      /* len(GameState.Players)                                      */
  uint8_t* alive;        /* Player.Alive                    (state.go:9)                */
  int32_t* army_count;   /* Player.ArmyCount                (state.go:10)               */
  int32_t* tile_count;   /* len(Player.OwnedTiles)                                      */
  int32_t* general_idx;  /* Player.GeneralIdx (highest listed general tile, -1 if none) */
} gvec_state_view;

typedef struct gvec_rollout_stats {
  int64_t env_steps;     /* turns actually advanced (live envs x turns)                 */
  int64_t aborted_turns; /* turns that returned a move-validation error     */
  int64_t games_finished;
  int64_t reserved;      /* always 0 (keeps the struct at 32 bytes)                        */
} gvec_rollout_stats;

/* ---- lifecycle ---------------------------------------------------------------- */

int32_t gvec_abi_version(void);
/* Fills the reference defaults (config.go:206-209, engine_initializer.go:118). */
int32_t gvec_config_default(gvec_config* cfg);
/* Replaces game.NewEngine for B engines (engine.go:62-71); boards arrive via gvec_reset. */
int32_t gvec_create(const gvec_config* cfg, gvec_handle** out);
int32_t gvec_destroy(gvec_handle* h);
/* One handle over several GPUs ("device list ... one handle may span several GPUs").  The B = cfg->num_envs
 * boards are split into num_devices contiguous shards whose sizes differ by at most one (shard i: base + (i < rem) envs
 * from i*base + min(i, rem), base = B / n, rem = B % n); shard i lives on devices[i] for the whole run (cfg->device is
 * ignored; a device may be listed more than once).  Boards are independent: no call moves board state between devices.
 * Every entry point that takes GVEC_MEM_HOST buffers works on the sharded handle exactly as on a plain one - the
 * caller's arrays are env-major over all B envs, every shard serves its slice, all devices at once - and the batch plays
 * the same games whatever the number of shards (a shard's agent / auto-reset / map-generator draws are keyed by the
 * env's index in the BATCH).  Entry points that take device pointers (GVEC_MEM_DEVICE, gvec_gym_*, gvec_export/import_
 * records, gvec_device_buffer, gvec_set_stream, gvec_experience_records) act on one device: use them on the child handle
 * gvec_shard returns (valid until the sharded handle is destroyed; never destroy it yourself).  Still not thread-safe:
 * one caller at a time per sharded handle (internally one worker thread per shard). */
int32_t gvec_create_sharded(const gvec_config* cfg, const int32_t* devices, int32_t num_devices, gvec_handle** out);
int32_t gvec_num_shards(const gvec_handle* h);   /* 0 for a plain handle */
int32_t gvec_shard(gvec_handle* h, int32_t i, gvec_handle** child, int32_t* env_begin, int32_t* num_envs, int32_t* device);
/* The one exchange step of the path on a sharded handle: every shard writes the compact experience records
 * (gvec_experience_records: needs gvec_experience_begin[_range] before the step) of ITS envs
 * [shard_env_begin, shard_env_begin + n) - indices inside the shard, the same slice on every device - and ships them to one
 * place: dst receives num_shards * n records, shard i's at record i * n; a record's env id is env_id_base + the env's
 * index in the batch.  mem = GVEC_MEM_HOST: dst is host memory, each device copies its slab straight to it over its own
 * PCIe link (what the Go StreamAggregator side, internal/grpc/gameserver/stream_aggregator.go:75-155, consumes);
 * mem = GVEC_MEM_DEVICE: dst is device memory on dst_device, the slabs travel GPU-to-GPU (hipMemcpyPeerAsync: xGMI on an
 * MI355X node) - for a learner that lives on a GPU.  Returns when every slab has landed. */
int32_t gvec_gather_experience_records(gvec_handle* h, int32_t shard_env_begin, int32_t n, int32_t env_id_base, int32_t mem,
                                       int32_t dst_device, void* dst);
/* Work is enqueued on this hipStream_t (default: the null stream). */
int32_t gvec_set_stream(gvec_handle* h, void* hip_stream);
int32_t gvec_synchronize(gvec_handle* h);
const char* gvec_last_error(void);

/* Page-locked host memory for the GVEC_MEM_HOST entry points: buffers obtained here are copied to / from the device at full
 * PCIe rate (a pageable buffer - a Go slice, a numpy array - goes through the driver's bounce buffers at a fraction of
 * it: 218 MB of legal masks per step at 262,144 boards take 18 ms pageable, 4 ms pinned).  Any host pointer is accepted by
 * every entry point; these are simply faster.  A cgo host passes them as unsafe.Pointer / slices over C memory. */
int32_t gvec_host_alloc(uint64_t bytes, void** out);
int32_t gvec_host_free(void* p);

/* sizes */
int32_t gvec_num_envs(const gvec_handle* h);
int32_t gvec_tile_stride(const gvec_handle* h);   /* max_width*max_height              */
int32_t gvec_mask_bytes(const gvec_handle* h);    /* bytes of one player's packed mask: 4 direction bit-planes (see gvec_step) */
int64_t gvec_state_bytes_per_env(const gvec_handle* h); /* bytes of one env's state record (gvec_export_records) */

/* ---- reset: EngineInitializer.Initialize minus mapgen -------------------------
 * (engine_initializer.go:113-143,218-225): upload boards, Turn=0, then the a15
 * pass: full stats (stats.go:33-63), full fog (visibility_optimized.go:33-53) if
 * fog_of_war, game-over check (engine.go:160-194).
 * env_ids NULL => envs 0..n-1.  owner must be in [-1, P). */
int32_t gvec_reset(gvec_handle* h, const int32_t* env_ids, int32_t n,
                   const int32_t* army, const int8_t* owner, const uint8_t* type,
                   const int32_t* width, const int32_t* height, const int32_t* players,
                   int32_t mem);

/* Deterministic on-device map generator with the algorithm and ratios of
 * internal/game/mapgen/generator.go:25-253 (Go math/rand is not reproducible
 * here: boards are keyed by (seed, env) through the library's own counter RNG).
 * Generates a board for every env (sizes per env: width/height/players arrays
 * [B] on host, or NULL = max sizes) and runs the reset pass. */
int32_t gvec_reset_generated(gvec_handle* h, uint64_t seed,
                             const int32_t* width, const int32_t* height,
                             const int32_t* players);
/* The same generator on Go's OWN math/rand: env i gets exactly the board
 *     mapgen.NewGenerator(mapgen.DefaultMapConfig(W, H, P), rand.New(rand.NewSource(seeds[i]))).GenerateMap()
 * builds (mapgen/generator.go:56-75) - i.e. the board game.NewEngine(ctx, GameConfig{Width, Height, Players, Rng:
 * rand.New(rand.NewSource(seeds[i]))}) starts from (engine_initializer.go:106-110) - then the reset pass.  Go's generator
 * (go 1.24: an additive lagged Fibonacci generator seeded through an LCG and a 607-word table) is restated from its
 * published algorithm, the table derived (scripts/gen_go_rand_cooked.py), and the whole pinned by the reference's own
 * seed-12345 tests (mapgen/generator_test.go:61-85, :396-455) through the oracle twin.  seeds: host int64 [B]; sizes as
 * gvec_reset_generated.  A Go host can therefore build B reference engines from seeds and this handle from the same seeds
 * and compare them from turn 0 (host/go/vecengine_diff_test.go). */
int32_t gvec_reset_go_seeded(gvec_handle* h, const int64_t* seeds, const int32_t* width, const int32_t* height,
                             const int32_t* players);
/* Pool of pre-generated, pre-initialised boards used by auto_reset: board j is
 * generated with key (seed, j) and sizes width[j]/height[j]/players[j] (host arrays
 * [pool_size], NULL = max sizes).  A finished env spends its next step being
 * re-dealt board mulhi(fmix32(env_key(seed, env) ^ episode*0x9E3779B1), pool_size). */
int32_t gvec_build_board_pool(gvec_handle* h, int32_t pool_size, uint64_t seed,
                              const int32_t* width, const int32_t* height,
                              const int32_t* players);

/* ---- the hot path ---------------------------------------------------------------
 * Engine.Step for every env (engine.go:75 -> turn_processor.go:29-77):
 * actions[B][max_players]; err[B] receives the sentinel of the first failing move
 * in PlayerID order (processor/action_processor.go:36-99), GVEC_ERR_GAME_OVER for
 * a finished env (turn_processor.go:95-113), else 0.
 * legal_bits (may be NULL): [B][max_players][mask_bytes], computed on the post-step
 * state.  One player's mask is FOUR DIRECTION BIT-PLANES of mask_bytes/4 bytes each,
 * d = 0 up, 1 right, 2 down, 3 left (rules/legal_moves.go:13-18,66): bit t (LSB
 * first) of plane d = Engine.GetLegalActionMask(p)[t*4 + d], t = y*W + x.  (Go's
 * []bool index order is a transpose of this: see INTEGRATION.md for the 3-line
 * unpack.)  ABI version 2: version 1 packed bit i = mask[i]. */
int32_t gvec_step(gvec_handle* h, const gvec_action* actions, int32_t* err,
                  uint8_t* legal_bits, int32_t mem);

/* Engine.GetLegalActionMask for all envs/players (engine.go:271-280). */
int32_t gvec_legal_mask(gvec_handle* h, uint8_t* legal_bits, int32_t mem);

/* Engine.ComputePlayerVisibility(player) for all envs
 * (visibility_optimized.go:166-195): visible/fog are [B][tile_stride] 0/1 bytes;
 * either may be NULL. */
int32_t gvec_player_visibility(gvec_handle* h, int32_t player,
                               uint8_t* visible, uint8_t* fog, int32_t mem);

/* Engine.GameState() / IsGameOver / GetWinner / GetChangedTiles /
 * GetVisibilityChangedTiles (engine.go:197-198,248-298) for envs
 * [env_begin, env_begin+n). */
int32_t gvec_read_state(gvec_handle* h, int32_t env_begin, int32_t n,
                        const gvec_state_view* view, int32_t mem);
/* Raw poke of engine state (what the reference's tests do by writing e.gs.*
 * directly, e.g. action_mask_test.go:62-68); no init pass is run. Planes that are
 * NULL keep their current contents. */
int32_t gvec_write_state(gvec_handle* h, int32_t env_begin, int32_t n,
                         const gvec_state_view* view, int32_t mem);

/* ---- synthetic random-agent rollouts (BASELINE.json metric) ---------------------
 * K turns for every env with the on-device random agent of SURVEY 8(d): per alive
 * player uniform choice over Engine.GetLegalActionMask, half with p=0.3, no-op
 * with p=0.1, `invalid_permille`/1000 deliberately unchecked moves (H5 stress);
 * counter RNG keyed by (seed, env, turn, player).  fused != 0 keeps board state in
 * registers/LDS across the K turns of one launch; fused == 0 issues one step launch
 * per turn (same results).  stats may be NULL. */
int32_t gvec_rollout(gvec_handle* h, int32_t turns, uint64_t seed,
                     int32_t invalid_permille, int32_t fused,
                     gvec_rollout_stats* stats);
/* The per-turn rollout (fused = 0, no statistics) for envs [env_begin, env_begin + n) only: the same launch over a slice
 * of the batch.  Disjoint slices may be stepped on different streams (gvec_set_stream before each call) and overlap on the
 * GPU - how a consumer steps a sampled slice together with its snapshot / record kernels beside the rest of the batch
 * (bench.py's gathering steps).  Stepping every env exactly once, slice by slice, equals one gvec_rollout turn. */
int32_t gvec_rollout_range(gvec_handle* h, int32_t env_begin, int32_t n, int32_t turns, uint64_t seed, int32_t invalid_permille);
/* The agent's mix: a player sits a turn out when (draw & 0xFFFF) < noop_per_65536 (default 6554,
 * p = 0.1) and moves half its army when (draw >> 16) < half_per_65536 (default 19661, p = 0.3).
 * (45875, 19661) are the rates of the reference's game.GenerateRandomActions (demo_helpers.go:20,44:
 * a player acts with p = 0.3, MoveAll with p = 0.7); (0, 0) always plays a full move, uniform over
 * the legal ones.  The default acts three times as often as the Go helper: more work per turn.
 * Applies to gvec_rollout and gvec_agent_actions. */
int32_t gvec_set_agent_mix(gvec_handle* h, int32_t noop_per_65536, int32_t half_per_65536);
/* Lifetime counters summed over every env of the handle (absolute values; gvec_rollout's stats are the
 * difference of two such reads): env_steps = engine turns actually PLAYED - a step that re-deals a finished
 * env, a frozen env or a GVEC_ACT_SKIP_ENV env plays none -, aborted_turns, games_finished.
 * One small reduction launch + an 24-byte read-back; synchronises the handle's stream. */
int32_t gvec_counters(gvec_handle* h, gvec_rollout_stats* out);
/* The HBM bytes ONE env-step of the hot path (gvec_step / per-turn gvec_rollout) must move BY CONSTRUCTION of the
 * resident layout, from the same constants the kernel is compiled with:
 * out4[0] read  = header + mutable and constant planes + narrow armies
 * out4[1] write = header + mutable planes + narrow armies
 * out4[2] mask  = the legal masks written when legal_bits / the agent is on (max_players * mask_bytes)
 * out4[3] extra = what an env in the rare forms adds on top (list planes both ways + the int32 army escape both ways)
 * bench.py prices roofline.frac with read + write + mask. */
int32_t gvec_step_traffic_bytes(const gvec_handle* h, int64_t* out4);
/* The agent alone: fills actions[B][max_players] for the current state/turn. */
int32_t gvec_agent_actions(gvec_handle* h, uint64_t seed, int32_t invalid_permille,
                           gvec_action* actions, int32_t mem);

/* ---- internal/experience side channel ------------------------------
 * gvec_experience_begin   = TurnProcessor.captureStateForExperience
 *                           (turn_processor.go:116-121: GameState.Clone before the step):
 *                           snapshots what the reward needs from the current state.
 * gvec_experience_rewards = CalculateReward(prev, cur, player) for every env / player
 *                           (internal/experience/rewards.go:40-85, DefaultRewardConfig
 *                           :23-37), prev = the snapshot, cur = the resident state;
 *                           rewards[B][max_players] float32, done[B] =
 *                           GameState.IsGameOver (state.go:73-82), may be NULL.  A board
 *                           that was re-dealt since the snapshot gets reward 0.
 * gvec_observe            = Serializer.StateToTensor(state, player)
 *                           (internal/experience/serializer.go:37-109): float32
 *                           [9][H][W] at index c*H*W + y*W + x inside a slot of
 *                           9*tile_stride floats; player >= 0: out[B][9*stride];
 *                           player = -1: out[B][max_players][9*stride].
 * gvec_serializer_mask    = Serializer.GenerateActionMask (serializer.go:112-176):
 *                           same packing as gvec_legal_mask but the serializer's
 *                           semantics (board owner, army >= 2, no Alive check) and
 *                           direction order 0 up, 1 down, 2 left, 3 right. */
int32_t gvec_experience_begin(gvec_handle* h);
/* The same snapshot for envs [env_begin, env_begin+n) only (a consumer that samples a slice of the batch). */
int32_t gvec_experience_begin_range(gvec_handle* h, int32_t env_begin, int32_t n);
int32_t gvec_experience_rewards(gvec_handle* h, float* rewards, uint8_t* done, int32_t mem);

/* ---- experience records: what a rank ships to the process feeding StreamAggregator ----
 * One compact record per env transition = everything SimpleCollector.OnStateTransition
 * (internal/experience/collector.go:30-98) puts into the experiencepb.Experience of every player that
 * acted, with the two [9][H][W] float tensors per player replaced by what they are functions of:
 *   dword 0 currState.Turn | 1 W | H<<8 | P<<16 | flags<<24 (1 done = currState.IsGameOver, 2 fog of war,
 *   4 valid: the env was stepped, not re-dealt, since the snapshot) | 2 acted bits (bit p: player p
 *   submitted an action) | 3 env id (env_id_base + env) |
 *   action[MP] int32 (Serializer.ActionToIndex on prevState.Board.W, serializer.go:179-198; -1 none) |
 *   reward[MP] float32 (CalculateReward, rewards.go:40-85) |
 *   bit-planes of fd dwords (bit t = tile y*W+x): prev own[MP], prev visible[MP], next own[MP],
 *   next visible[MP], general, city, mountain |
 *   Serializer.GenerateActionMask(prevState, p) as [MP][4][fd] direction planes (0 up, 1 down, 2 left, 3 right) |
 *   prev armies, next armies: uint16[NS*64], tile t at halfword t, saturated to [0, 65535] - exact for
 *   StateToTensor, which clamps army/1000 at 1 (serializer.go:82-85).
 * MP / fd / NS come from gvec_experience_record_layout: out8 = {dwords per record, MP, fd, NS, max_players,
 * tile_stride, 0, 0}.  3,660 bytes at 20x20 4P instead of 115 KB of tensors; the consumer expands
 * (generalsreinforcementlearning_amd/experience.py: decode_records).
 * gvec_experience_records needs the snapshot of gvec_experience_begin[_range] taken BEFORE the step and the
 * actions[B][max_players] that were played (NULL = the handle's action buffer: the last host-mode
 * gvec_step, or the device agent's moves when gvec_record_agent_actions is on); it writes n records to
 * dst_device on the handle's stream. */
int32_t gvec_experience_record_layout(gvec_handle* h, int32_t* out8);
int32_t gvec_experience_record_bytes(gvec_handle* h);
int32_t gvec_experience_records(gvec_handle* h, const gvec_action* actions, int32_t mem, int32_t env_begin,
                                int32_t n, int32_t env_id_base, void* dst_device);
/* The CONSUMER side of the exchange (the GPU the records were gathered to, e.g. rank 0 feeding StreamAggregator or a
 * learner): expands n compact records (device memory on `device`) into the fields of the experiencepb.Experience messages
 * SimpleCollector.OnStateTransition builds (collector.go:41-75), one slot per (record, player), [n][MP] record-major,
 * MP = layout8[1] (layout8 as gvec_experience_record_layout fills it - the sender's; no engine handle is needed here):
 *   state, next_state  float32 [n][MP][9 * stride]: StateToTensor(prevState / currState, p), index c*H*W + y*W + x with
 *                      the record's own W, H at the start of the slot (stride = layout8[5]);
 *   action_mask        uint8   [n][MP][4 * stride]: GenerateActionMask(prevState, p) as 0/1 bytes, index t*4 + d,
 *                      d = 0 up, 1 down, 2 left, 3 right (the []bool of serializer.go:112-176);
 *   meta               int32   [n][MP][8]: present (1: the record is valid and player p acted - only then the slot
 *                      holds an experience, else it is zeroed), env id, player id, currState.Turn, action index, reward
 *                      (float32 bits), done, W | H << 8.
 * Enqueued on hip_stream (may be NULL) of `device`; device pointers only.  experience.py: expand_records_device. */
int32_t gvec_expand_experience_records(int32_t device, void* hip_stream, const int32_t* layout8, const void* records, int32_t n,
                                       float* state, float* next_state, uint8_t* action_mask, int32_t* meta);
/* on != 0: per-turn rollouts (gvec_rollout fused = 0) store the agent's moves in the handle's action buffer and the
 * per-env error codes (the sentinel of the first failing move, 0, or GVEC_ERR_GAME_OVER) in its err buffer. */
int32_t gvec_record_agent_actions(gvec_handle* h, int32_t on);
int32_t gvec_observe(gvec_handle* h, int32_t player, float* out, int32_t mem);
int32_t gvec_serializer_mask(gvec_handle* h, uint8_t* bits, int32_t mem);

/* ---- the gRPC surface's per-turn broadcast --------------------------------------------------
 * gameInstance.createStreamUpdate (internal/grpc/gameserver/server.go:632-777) decides per env and player stream: when
 * 0 < |ChangedTiles| + |VisibilityChangedTiles| < W*H/5 the update is a GameStateDelta - the tiles of either set with the
 * proto's fog rules for that player (:664-689) plus every PlayerState - else the full GameState.  gvec_stream_deltas
 * makes that decision and builds the delta's tile updates on the device, so that a turn's broadcast reads back a few
 * eight-byte updates per env instead of the board:
 *   kind[B]    1 = delta, 2 = the server sends convertGameStateToProto's full state (gvec_read_state + gvec_player_visibility)
 *   count[B]   tile updates of the delta (0 for kind 2)
 *   updates    [B][cap] uint64, cap = gvec_stream_delta_cap(h) (= max(1, tile_stride / 5): a delta has fewer than N/5):
 *              bits 0-15 tile index y*W + x | 16-17 Tile.Type (core numbering, after the fog rules) | 18 visible |
 *              19 fog_of_war | 20-23 owner + 1 (0: -1, i.e. neutral or withheld) | 32-63 army (int32, 0 when withheld);
 *              the ChangedTiles in ascending order, then the tiles only in VisibilityChangedTiles (Go's map order is
 *              unspecified).  PlayerUpdates come from gvec_read_state's per-player fields (a few bytes per env).
 * GVEC_MEM_HOST or GVEC_MEM_DEVICE. */
int32_t gvec_stream_delta_cap(const gvec_handle* h);
int32_t gvec_stream_deltas(gvec_handle* h, int32_t player, uint8_t* kind, int32_t* count, uint64_t* updates, int32_t mem);
/* The same updates as ONE stream (host memory): env e's updates are updates[offset[e] .. offset[e + 1]), offset has B + 1
 * entries, *total = offset[B].  Only the updates that exist cross PCIe - about 11 per env-turn at 20x20 4P, 25 MB for
 * 262,144 boards instead of the 168 MB of the fixed-stride form or the 1 GB of the boards.  capacity = the entries
 * `updates` can hold: GVEC_E_RANGE (with *total set) when it is too small; B * gvec_stream_delta_cap(h) always suffices
 * (B * tile_stride with full_tiles).  full_tiles != 0: an env of kind 2 contributes ALL its W*H tiles in the same packed
 * form, ascending - the board of convertGameStateToProto's full state with the player's fog rules applied
 * (server.go:556-582) - so that a broadcast never reads a board back, growth turns included.  Plain handles only. */
int32_t gvec_stream_deltas_packed(gvec_handle* h, int32_t player, int32_t full_tiles, uint8_t* kind, int64_t* offset, uint64_t* updates,
                                  int64_t capacity, int64_t* total);

/* ---- python/generals_gym on the device ------------------------------------------
 * What GeneralsEnv builds on the client from the GameState proto of its player token, computed straight
 * from the resident state with the proto's fog rules (server.go:556-582) applied in the kernel.  Every
 * pointer is a DEVICE pointer (e.g. a torch tensor); work is enqueued on the handle's stream.
 * gvec_gym_observe: obs [B][9][tile_stride] float32 = GeneralsEnv._get_observation (generals_env.py:291-342:
 *   visible, ownership 0.5 / 1.0, log(army+1)/10, one-hot type, turn_count/max_turns, zeros), tile index
 *   y*W + x; mask [B][tile_stride*5] 0/1 bytes = _get_valid_actions_mask (:344-387), index tile*5 + {up, right,
 *   down, left, half}; reward [B] float64 = _calculate_reward (:499-561) against the player stats stored by
 *   the PREVIOUS gvec_gym_observe call of this handle (the call after a reset yields a meaningless reward);
 *   done = Engine.IsGameOver, winner = Engine.GetWinner.  reward / done / winner may be NULL.
 * gvec_gym_actions: GeneralsEnv.step's action handling (:226-259, :389-441) for `player`: decodes
 *   gym_actions[B] (indices into Discrete(board_size*5)) against `mask` (the last gvec_gym_observe's), writes
 *   the player's move into actions[B][max_players] (the other players' slots are kept: fill them first, e.g.
 *   with gvec_agent_actions), marks an env whose action is refused GVEC_ACT_SKIP_ENV and a `resetting` env
 *   GVEC_ACT_RESET_ENV; played / invalid / error [B] 0/1 outputs (any may be NULL). */
int32_t gvec_gym_observe(gvec_handle* h, int32_t player, const int64_t* turn_count, int32_t max_turns,
                         float* obs, uint8_t* mask, double* reward, uint8_t* done, int8_t* winner);
/* gvec_gym_observe plus the bookkeeping GeneralsEnv.step wraps around it (generals_env.py:226-259), in the same
 * kernel: per env, with rs = resetting[env] (this step re-dealt it) and pl = played[env] (gvec_gym_actions accepted
 * the action): turn_count := rs ? 0 : turn_count + pl (in place, and copied to turn_out); the observation uses the new
 * count; reward := rs ? 0 : (pl ? _calculate_reward : -0.1); terminated := game over & pl & !rs; truncated :=
 * turn_count >= max_turns & pl & !rs; needs_reset := terminated | truncated; winner := terminated ? GetWinner : -1.
 * Outputs other than obs / mask may be NULL.  One launch instead of one launch and a dozen elementwise tensor ops. */
int32_t gvec_gym_finish_step(gvec_handle* h, int32_t player, int64_t* turn_count, int32_t max_turns,
                             const uint8_t* resetting, const uint8_t* played, float* obs, uint8_t* mask, double* reward,
                             uint8_t* terminated, uint8_t* truncated, int8_t* winner, uint8_t* needs_reset,
                             int64_t* turn_out);
int32_t gvec_gym_actions(gvec_handle* h, int32_t player, const int64_t* gym_actions, const uint8_t* mask,
                         const uint8_t* resetting, gvec_action* actions, uint8_t* played, uint8_t* invalid,
                         uint8_t* error);

/* GeneralsEnv.step (generals_env.py:226-259) for every env in ONE launch: exactly
 *   gvec_agent_actions(agent_seed, 0) -> gvec_gym_actions(gym_actions, the last observation's mask, resetting) ->
 *   gvec_step(device actions, no err / legal output) -> gvec_gym_finish_step(resetting, played, ...)
 * with the same outputs bit for bit, without the intermediate buffers: the learner's action is decoded against the
 * valid-action mask of the resident state (recomputed on the fly), the other players move like the on-device agent, the
 * turn is played and observation / mask / reward / flags of the new state are written while the board is in registers.
 * `resetting` = the needs_reset output of the previous call (zeros after a reset).  Needs auto_reset and a board pool
 * (GVEC_E_INVALID otherwise: a re-deal is how an episode ends here).  obs / mask / turn_count / gym_actions / resetting
 * are required, every other output may be NULL.  Device pointers only; enqueued on the handle's stream. */
int32_t gvec_gym_step(gvec_handle* h, int32_t player, uint64_t agent_seed, const int64_t* gym_actions, const uint8_t* resetting,
                      int64_t* turn_count, int32_t max_turns, float* obs, uint8_t* mask, double* reward, uint8_t* terminated,
                      uint8_t* truncated, int8_t* winner, uint8_t* needs_reset, int64_t* turn_out, uint8_t* played,
                      uint8_t* invalid, uint8_t* error);

/* The collection loop around GeneralsEnv.step, resident on the device: one iteration of every worker's
 * ParallelEnvPool._run_episode (python/generals_gym/vector_env.py:164-192) plus ReplayBuffer.push
 * (python/generals_gym/replay_buffer.py:31-36) for num_envs workers, on the outputs of one gvec_gym_step, without a byte
 * crossing PCIe.  Per worker w, with live = !was_reset[w] (this step was not the worker's env.reset()):
 *   live:  the transition (state[w], action[w], reward[w], next_state[w], terminated[w] | truncated[w]) goes to the ring
 *          (workers in ascending order from the ring's cursor; the oldest transitions are overwritten once `capacity` is
 *          reached - deque(maxlen) semantics); episode_reward[w] += reward[w]; episode_length[w] += 1;
 *   over = live & (done | episode_length[w] >= max_steps_per_episode): (episode_reward, episode_length, w) is appended
 *          to the result log (vector_env.py:189-190; entries beyond result_capacity are counted, not kept), the two
 *          accumulators are zeroed, and needs_reset[w] is raised when the episode was cut at the length limit (so that
 *          the NEXT gvec_gym_step re-deals the env - the worker's next env.reset()).
 * ring_counters[4]  = {cursor, size, total pushed, 0};  pool_counters[4] = {episodes, results held, results dropped, 0}
 * (int64, device memory; zero them to start; "results held" is reset by the consumer after it has read the log).
 * scratch: gvec_pool_collect_scratch_bytes(num_envs) bytes, 16-byte aligned, contents of no interest to the caller.  Every
 * pointer is device memory on `device`; capacity >= num_envs.  Three launches on hip_stream: per-worker flags, one
 * workgroup's prefix counts (which also moves the counters on), the copy (a row of the ring is obs_floats floats, moved
 * by one to eight wavefronts). */
typedef struct gvec_collect_args {
  int32_t num_envs, obs_floats, max_steps_per_episode, reserved;
  int64_t capacity, result_capacity;
  const float* state; const float* next_state; const int64_t* action; const double* reward;
  const uint8_t* terminated; const uint8_t* truncated; const uint8_t* was_reset;
  uint8_t* needs_reset;
  float* ring_state; float* ring_next_state; int64_t* ring_action; double* ring_reward; uint8_t* ring_done;
  int64_t* ring_counters;
  double* episode_reward; int64_t* episode_length;
  double* result_reward; int32_t* result_length; int32_t* result_worker;
  int64_t* pool_counters;
  void* scratch;
} gvec_collect_args;
int32_t gvec_pool_collect(int32_t device, void* hip_stream, const gvec_collect_args* args);
uint64_t gvec_pool_collect_scratch_bytes(int32_t num_envs);

/* ---- experience gather support ---------------------------------------
 * Writes the compact state records of envs [env_begin, env_begin+n) into a device
 * buffer (e.g. a torch tensor handed to RCCL) as a slab [n] headers | [n] plane blocks |
 * [n] int32 army blocks: gvec_state_bytes_per_env() bytes per env.  Armies are always
 * int32 in a record, whatever form the env is stored in.  gvec_import_records checks
 * every record header against the handle's limits (board size, player count, reciprocal)
 * on the device and fails with GVEC_E_BOARD - that env left untouched - on a mismatch. */
int32_t gvec_export_records(gvec_handle* h, int32_t env_begin, int32_t n,
                            void* dst_device);
int32_t gvec_import_records(gvec_handle* h, int32_t env_begin, int32_t n,
                            const void* src_device);
/* Zero-copy access for device consumers (torch-ROCm): the handle's resident device
 * arrays.  which: 0 header [B][24] u32, 1 bit-planes (the OwnedTiles planes at the end of an env's block are
 * meaningful only while its header flag bit 7 is set; otherwise the lists are the ownership planes - DESIGN.md
 * section 3), 2 armies (narrow form: u16, two 64-tile slots
 * interleaved per dword; authoritative for every env whose header flag bit 2 is clear - see
 * DESIGN.md section 3), 6 armies (wide form: int32, authoritative for the flagged envs), 3 legal masks
 * [B][max_players][mask_bytes], 4 actions [B][max_players], 5 err [B].  Consumers that want plain
 * per-tile planes use gvec_read_state with GVEC_MEM_DEVICE instead. */
#define GVEC_BUF_HEADER  0
#define GVEC_BUF_ROWS    1
#define GVEC_BUF_ARMY    2
#define GVEC_BUF_LEGAL   3
#define GVEC_BUF_ACTIONS 4
#define GVEC_BUF_ERR     5
#define GVEC_BUF_ARMY_WIDE 6
void*   gvec_device_buffer(gvec_handle* h, int32_t which);
/* A byte range of one of the handle's small per-env buffers - GVEC_BUF_HEADER, _LEGAL, _ACTIONS (what the device agent
 * played, while gvec_record_agent_actions is on) or _ERR (the per-env codes of the last gvec_step in host mode or, while
 * recording is on, of the last per-turn gvec_rollout launch) - copied to host memory; synchronises the handle's stream. */
int32_t gvec_read_buffer(gvec_handle* h, int32_t which, uint64_t byte_offset, uint64_t bytes, void* host_dst);

/* Runs the on-device self-test of the wave primitives (DPP shifts, scans,
 * bpermute); 0 = pass.  Used by smoke tests on new driver / hardware revisions. */
int32_t gvec_selftest(int32_t device);

#ifdef __cplusplus
}
#endif
#endif /* GENERALS_VEC_H */
