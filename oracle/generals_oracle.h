/*
 * generals_oracle.h — CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement of the reference Go turn engine
 * (mitchelldurbincs/GeneralsReinforcementLearning, internal/game +
 * internal/game/core + processor + rules), kept deliberately close to the Go
 * data structures: AoS tiles, ordered Player.OwnedTiles lists, ChangedTiles /
 * VisibilityChangedTiles sets.  Each function cites the Go file:line it follows.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  The product (generalsreinforcementlearning_amd/csrc) never
 * links or calls it.
 *
 * Parity pinning: the Go reference cannot be compiled here (no Go toolchain in
 * the image), so this restatement is pinned by the known-answer vectors held in
 * the reference's own *_test.go files, transcribed under tests/golden/ (see
 * tests/golden/README.md for the file:line of every vector).  Fog of war and the
 * H4-H8 hazards of SURVEY.md have no asserting test in the reference: for those
 * the oracle rests on line-by-line restatement only ("parity unpinned" for fog by
 * the reference's own tests).  Fog has two further checks of its own
 * (tests/test_fog_cross_checks.py): the reference's LEGACY twin of the update
 * (visibility.go:19-144) is restated separately below and must agree with the
 * optimized restatement on every turn, and a restatement-free property (visibility
 * = 3x3 dilation of last turn's lists while lists match the board).
 */
#ifndef GENERALS_ORACLE_H
#define GENERALS_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* core/board.go:20-26 */
enum { ORA_TILE_NORMAL = 0, ORA_TILE_GENERAL = 1, ORA_TILE_CITY = 2, ORA_TILE_MOUNTAIN = 3, ORA_NEUTRAL = -1 };
/* core/errors.go:8-17, numbered as proto/common/v1/common.proto:39-48 */
enum { ORA_OK = 0, ORA_ERR_INVALID_COORDINATES = 1, ORA_ERR_NOT_ADJACENT = 2, ORA_ERR_NOT_OWNED = 3,
       ORA_ERR_INSUFFICIENT_ARMY = 4, ORA_ERR_GAME_OVER = 5, ORA_ERR_INVALID_PLAYER = 6,
       ORA_ERR_MOVE_TO_SELF = 7, ORA_ERR_TARGET_IS_MOUNTAIN = 8 };

/* core/board.go:7-13 */
typedef struct ora_tile {
  int32_t  owner;
  int64_t  army;      /* Go int is 64-bit */
  int32_t  type;
  uint32_t visible;    /* VisibleBitfield */
  uint32_t discovered; /* DiscoveredBitfield (stays 0 on the optimized path) */
} ora_tile;

/* core/board.go:15-18 */
typedef struct ora_board { int32_t w, h; ora_tile* t; } ora_board;

/* core/action.go:23-36 (From/To Coordinate fields are never set on the gRPC path,
 * converters.go:116-123, and are not modelled) */
typedef struct ora_move {
  int32_t player_id, from_x, from_y, to_x, to_y, move_all;
} ora_move;

/* core/movement.go:7-17 */
typedef struct ora_capture {
  int32_t x, y, tile_type, capturing_player, previous_owner;
  int64_t previous_army;
} ora_capture;

/* core/movement.go:93-96 */
typedef struct ora_elimination { int32_t eliminated, new_owner; } ora_elimination;

/* ---- core package ---- */
ora_board* ora_board_new(int32_t w, int32_t h);                 /* core/board.go:97-106 */
void       ora_board_free(ora_board* b);
ora_tile*  ora_board_tile(ora_board* b, int32_t idx);
int32_t    ora_validate(const ora_board* b, const ora_move* m, int32_t player_id); /* core/action.go:56-105 */
/* core/movement.go:23-89; changed may be NULL (bitmap of w*h bytes); returns error code;
 * *captured = 1 and *cap filled on capture */
int32_t    ora_apply_move(ora_board* b, const ora_move* m, uint8_t* changed, ora_capture* cap, int32_t* captured);
int32_t    ora_process_captures(const ora_capture* caps, int32_t n, ora_elimination* out); /* core/movement.go:100-118 */
/* Tile.SetVisible / IsVisibleTo (core/board.go:46-64) */
void       ora_tile_set_visible(ora_tile* t, int32_t player, int32_t visible);
int32_t    ora_tile_is_visible_to(const ora_tile* t, int32_t player);

/* ---- game package: one Engine ---- */
typedef struct ora_engine ora_engine;

typedef struct ora_params {
  int32_t fog_of_war;              /* GameState.FogOfWarEnabled */
  int32_t prod_general, prod_city, prod_normal, normal_growth_interval; /* config.go:206-209 */
} ora_params;
void ora_params_default(ora_params* p);

/* initializeGameState + initializePlayers (engine_initializer.go:113-143) on a
 * caller-supplied board (owner/army/type planes, row-major), WITHOUT the initial
 * setup pass; ora_engine_initial_setup runs performInitialSetup (:218-225). */
ora_engine* ora_engine_new(int32_t w, int32_t h, int32_t players, const ora_params* params,
                           const int32_t* army, const int8_t* owner, const uint8_t* type);
void        ora_engine_free(ora_engine* e);
void        ora_engine_initial_setup(ora_engine* e);
/* Engine.Step (engine.go:75 -> turn_processor.go:29-77); actions need not be sorted.
 * Returns 0 or the sentinel code of the returned error. */
int32_t     ora_engine_step(ora_engine* e, const ora_move* actions, int32_t n);
void        ora_engine_update_player_stats(ora_engine* e);      /* stats.go:8-30 */
void        ora_engine_update_fog(ora_engine* e);               /* visibility.go:11-16 -> visibility_optimized.go:16-30 */
/* visibility.go:19-144: the legacy twin of the fog update, restated independently (test-only cross-check:
 * identical VisibleBitfield, plus the DiscoveredBitfield the optimized path never sets - SURVEY H9) */
void        ora_engine_update_fog_legacy(ora_engine* e);
void        ora_engine_process_production(ora_engine* e);       /* production_manager.go:26-73 with gs.Turn */
void        ora_engine_check_game_over(ora_engine* e);          /* engine.go:160-194 */
/* Engine.GetLegalActionMask (engine.go:271-280 -> rules/legal_moves.go:19-73); mask has w*h*4 bytes */
void        ora_engine_legal_mask(const ora_engine* e, int32_t player, uint8_t* mask);
/* Engine.ComputePlayerVisibilityOptimized (visibility_optimized.go:166-195) */
void        ora_engine_player_visibility(const ora_engine* e, int32_t player, uint8_t* visible, uint8_t* fog);
int32_t     ora_engine_is_game_over(const ora_engine* e);       /* engine.go:198 */
int32_t     ora_engine_winner(const ora_engine* e);             /* engine.go:248-263 */
/* raw access (what the Go tests do with e.gs.*) */
ora_board*  ora_engine_board(ora_engine* e);
int32_t     ora_engine_turn(const ora_engine* e);
void        ora_engine_set_turn(ora_engine* e, int32_t turn);
void        ora_engine_set_game_over(ora_engine* e, int32_t v);
void        ora_engine_set_fog(ora_engine* e, int32_t enabled);
int32_t     ora_engine_num_players(const ora_engine* e);
int32_t     ora_player_alive(const ora_engine* e, int32_t p);
void        ora_player_set_alive(ora_engine* e, int32_t p, int32_t alive);
int64_t     ora_player_army_count(const ora_engine* e, int32_t p);
int32_t     ora_player_general_idx(const ora_engine* e, int32_t p);
void        ora_player_set_general_idx(ora_engine* e, int32_t p, int32_t idx);
int32_t     ora_player_num_owned(const ora_engine* e, int32_t p);
const int32_t* ora_player_owned(const ora_engine* e, int32_t p);
void        ora_player_set_owned(ora_engine* e, int32_t p, const int32_t* tiles, int32_t n);
int32_t     ora_engine_changed_count(const ora_engine* e);
int32_t     ora_engine_vis_changed_count(const ora_engine* e);
const uint8_t* ora_engine_changed(const ora_engine* e);       /* w*h bytes 0/1 */
const uint8_t* ora_engine_vis_changed(const ora_engine* e);

/* ---- batch of engines behind the same plane formats as include/generals_vec.h ---- */
typedef struct ora_batch ora_batch;
typedef struct ora_action8 { int8_t from_x, from_y, to_x, to_y; uint8_t flags; uint8_t reserved[3]; } ora_action8;

ora_batch* ora_batch_new(int32_t num_envs, int32_t max_w, int32_t max_h, int32_t max_p, const ora_params* params);
void       ora_batch_free(ora_batch* b);
/* = gvec_reset; env_ids NULL => 0..n-1 */
int32_t    ora_batch_reset(ora_batch* b, const int32_t* env_ids, int32_t n, const int32_t* army, const int8_t* owner,
                           const uint8_t* type, const int32_t* w, const int32_t* h, const int32_t* p);
/* = gvec_step; threads <= 1 runs serially, else OpenMP over envs. legal_bits may be NULL. */
int32_t    ora_batch_step(ora_batch* b, const ora_action8* actions, int32_t* err, uint8_t* legal_bits, int32_t threads);
int32_t    ora_batch_legal_mask(ora_batch* b, uint8_t* legal_bits, int32_t threads);
ora_engine* ora_batch_engine(ora_batch* b, int32_t env);
/* = gvec_read_state planes (any pointer may be NULL) */
typedef struct ora_state_view {
  int32_t* army; int8_t* owner; uint8_t* type; uint8_t* visible; int8_t* listed; uint8_t* changed; uint8_t* vis_changed;
  int32_t* turn; uint8_t* done; int8_t* winner; int32_t* width; int32_t* height; int32_t* players;
  uint8_t* alive; int32_t* army_count; int32_t* tile_count; int32_t* general_idx;
} ora_state_view;
int32_t    ora_batch_read_state(ora_batch* b, int32_t env_begin, int32_t n, const ora_state_view* v);
int32_t    ora_batch_write_state(ora_batch* b, int32_t env_begin, int32_t n, const ora_state_view* v);

/* what the LEGACY fog update would leave after the next turn's initializeTurn, for every env (batch untouched) */
int32_t    ora_batch_next_fog_legacy(ora_batch* b, uint8_t* visible /*[B][stride]*/, uint8_t* discovered /*[B][stride]*/);

/* ---- internal/experience: serializer + rewards ---- */
/* GameState.Clone (state.go:37-70) */
ora_engine* ora_engine_clone(const ora_engine* e);
/* Serializer.StateToTensor (internal/experience/serializer.go:37-109): out[9*w*h], index c*h*w + y*w + x */
void       ora_state_to_tensor(const ora_engine* e, int32_t player, float* out);
/* Serializer.GenerateActionMask (serializer.go:112-176): mask[w*h*4], d = 0 up, 1 down, 2 left, 3 right */
void       ora_serializer_mask(const ora_engine* e, int32_t player, uint8_t* mask);
/* CalculateReward (internal/experience/rewards.go:40-85) with DefaultRewardConfig (:23-37) */
float      ora_calculate_reward(const ora_engine* prev, const ora_engine* cur, int32_t player);
float      ora_army_advantage(const ora_engine* e, int32_t player);               /* rewards.go:153-175 */
void       ora_city_changes(const ora_engine* prev, const ora_engine* cur, int32_t player, int32_t* gained, int32_t* lost); /* :110-129 */
/* batch forms: begin = TurnProcessor.captureStateForExperience (turn_processor.go:116-121) */
int32_t    ora_batch_experience_begin(ora_batch* b);
int32_t    ora_batch_rewards(ora_batch* b, float* rewards /*[B][max_p]*/, uint8_t* done /*[B] or NULL*/);
int32_t    ora_batch_observe(ora_batch* b, int32_t player, float* out /*[B][9*stride]*/);
int32_t    ora_mask_bytes(int32_t stride);  /* = gvec_mask_bytes(): 4 direction bit-planes per player */
int32_t    ora_batch_serializer_mask(ora_batch* b, uint8_t* bits /*[B][max_p][mask_bytes]*/);

/* ---- synthetic inputs (the build's own spec; no reference counterpart
 * is reproducible here because Go math/rand is absent) ---- */
uint32_t   ora_fmix32(uint32_t h);
uint32_t   ora_amix(uint32_t x);      /* the random agent's mixer (24-bit multiplies) */
/* random agent: fills actions[num_envs][max_p] for the current state */
int32_t    ora_batch_set_agent_mix(ora_batch* b, int32_t noop_per_65536, int32_t half_per_65536);
int32_t    ora_batch_agent_actions(ora_batch* b, uint64_t seed, int32_t invalid_permille, ora_action8* actions, int32_t threads);
/* map generator (algorithm of mapgen/generator.go:64-253, own counter RNG) */
/* Go's math/rand (rand.New(rand.NewSource(seed))) and the reference's generator on it: see generals_oracle.c */
typedef struct ora_gorand ora_gorand;
ora_gorand* ora_gorand_new(int64_t seed);
void       ora_gorand_free(ora_gorand* r);
void       ora_gorand_seed(ora_gorand* r, int64_t seed);
int64_t    ora_gorand_int63(ora_gorand* r);
int32_t    ora_gorand_intn(ora_gorand* r, int32_t n);
int32_t    ora_mapgen_go(int64_t seed, int32_t w, int32_t h, int32_t players, const int32_t* cfg7 /*NULL: DefaultMapConfig*/,
                         int32_t* army, int8_t* owner, uint8_t* type);
int32_t    ora_mapgen(uint64_t seed, int32_t env, int32_t w, int32_t h, int32_t players,
                      int32_t* army, int8_t* owner, uint8_t* type);
/* auto-reset pool (= gvec_build_board_pool): board j = ora_mapgen(seed, j, w[j], h[j], p[j]);
 * a finished env spends its next step being re-dealt board
 * mulhi(fmix32(env_key(seed, env) ^ episode*0x9E3779B1), pool_size).  NULL sizes = max sizes. */
int32_t    ora_batch_set_pool(ora_batch* b, int32_t pool_size, uint64_t seed, const int32_t* w, const int32_t* h, const int32_t* p);
/* K turns of agent+step for every env; returns env-steps advanced. */
int64_t    ora_batch_rollout(ora_batch* b, int32_t turns, uint64_t seed, int32_t invalid_permille, int32_t threads);
int64_t    ora_batch_rollout_masks(ora_batch* b, int32_t turns, uint64_t seed, int32_t invalid_permille, int32_t threads,
                                   uint8_t* legal_bits /*[B][max_p][mask_bytes] or NULL*/);

#ifdef __cplusplus
}
#endif
#endif
