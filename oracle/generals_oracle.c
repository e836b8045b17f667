/*
 * generals_oracle.c — CPU ORACLE (test infrastructure, NOT product code).
 * See generals_oracle.h for scope, pinning and who may load this.
 *
 * Every function restates one Go function of the reference and cites it as
 * file:line relative to /root/reference/internal/game/.  The Go data structures
 * are kept (AoS tiles, ordered OwnedTiles lists, sets) on purpose: the device
 * code uses bit-planes instead, so this file checks that re-design, not itself.
 *
 * Go map iteration order is random; wherever the reference ranges over a map
 * (ChangedTiles, VisibilityChangedTiles, tempTileOwnership, tempAffectedPlayers)
 * this file iterates in ascending key order.  The only observable this can
 * change is the ORDER of OwnedTiles and, through it, which general tile
 * Player.GeneralIdx names when a player holds two or more; every
 * other output is order-independent.
 */
#include "generals_oracle.h"
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------- */
/* small containers                                                          */
/* ------------------------------------------------------------------------- */
typedef struct ilist { int32_t* v; int32_t n, cap; } ilist;

static void ilist_push(ilist* l, int32_t x) {
  if (l->n == l->cap) {
    l->cap = l->cap ? l->cap * 2 : 64;
    l->v = (int32_t*)realloc(l->v, sizeof(int32_t) * (size_t)l->cap);
  }
  l->v[l->n++] = x;
}

/* a Go map[int]struct{} keyed by tile index */
typedef struct tset { uint8_t* bits; int32_t count, n; } tset;
static void tset_init(tset* s, int32_t n) { s->bits = (uint8_t*)calloc((size_t)n, 1); s->count = 0; s->n = n; }
static void tset_add(tset* s, int32_t i) { if (!s->bits[i]) { s->bits[i] = 1; s->count++; } }
static void tset_clear(tset* s) { memset(s->bits, 0, (size_t)s->n); s->count = 0; }

/* ------------------------------------------------------------------------- */
/* core/board.go                                                             */
/* ------------------------------------------------------------------------- */
ora_board* ora_board_new(int32_t w, int32_t h) { /* core/board.go:97-106 */
  ora_board* b = (ora_board*)calloc(1, sizeof(ora_board));
  b->w = w; b->h = h;
  b->t = (ora_tile*)calloc((size_t)(w * h), sizeof(ora_tile));
  for (int i = 0; i < w * h; i++) { b->t[i].owner = ORA_NEUTRAL; b->t[i].type = ORA_TILE_NORMAL; }
  return b;
}
void ora_board_free(ora_board* b) { if (b) { free(b->t); free(b); } }
ora_tile* ora_board_tile(ora_board* b, int32_t idx) { return &b->t[idx]; }

static inline int board_idx(const ora_board* b, int x, int y) { return y * b->w + x; }          /* :108 */
static inline int in_bounds(const ora_board* b, int x, int y) { return x >= 0 && x < b->w && y >= 0 && y < b->h; } /* :112-114 */

void ora_tile_set_visible(ora_tile* t, int32_t p, int32_t visible) { /* core/board.go:53-64 */
  if (p < 0 || p >= 32) return;
  if (visible) { t->visible |= (1u << (unsigned)p); t->discovered |= (1u << (unsigned)p); }
  else t->visible &= ~(1u << (unsigned)p);
}
int32_t ora_tile_is_visible_to(const ora_tile* t, int32_t p) { /* core/board.go:46-51 */
  if (p < 0 || p >= 32) return 0;
  return (t->visible & (1u << (unsigned)p)) != 0;
}

/* ------------------------------------------------------------------------- */
/* core/action.go:56-105  MoveAction.Validate                                */
/* ------------------------------------------------------------------------- */
int32_t ora_validate(const ora_board* b, const ora_move* m, int32_t player_id) {
  if (!in_bounds(b, m->from_x, m->from_y)) return ORA_ERR_INVALID_COORDINATES;      /* :58-60 */
  if (!in_bounds(b, m->to_x, m->to_y)) return ORA_ERR_INVALID_COORDINATES;          /* :62-64 */
  if (m->from_x == m->to_x && m->from_y == m->to_y) return ORA_ERR_MOVE_TO_SELF;    /* :67-69 */
  { /* :72-76, Coordinate.IsAdjacentTo core/coordinate.go:47-53 */
    int dx = m->from_x - m->to_x, dy = m->from_y - m->to_y;
    int adj = (dx == 0 && (dy == 1 || dy == -1)) || (dy == 0 && (dx == 1 || dx == -1));
    if (!adj) return ORA_ERR_NOT_ADJACENT;
  }
  const ora_tile* from = &b->t[board_idx(b, m->from_x, m->from_y)];
  if (from->owner != player_id) return ORA_ERR_NOT_OWNED;                           /* :82-84 */
  if (from->army <= 1) return ORA_ERR_INSUFFICIENT_ARMY;                            /* :87-89 */
  const ora_tile* to = &b->t[board_idx(b, m->to_x, m->to_y)];
  if (to->type == ORA_TILE_MOUNTAIN) return ORA_ERR_TARGET_IS_MOUNTAIN;             /* :96-98 */
  return ORA_OK;
}

/* ------------------------------------------------------------------------- */
/* core/movement.go:23-89  ApplyMoveAction                                   */
/* ------------------------------------------------------------------------- */
static int32_t apply_move(ora_board* b, const ora_move* m, tset* changed_set, uint8_t* changed_bytes,
                          ora_capture* cap, int32_t* captured) {
  *captured = 0;
  int32_t err = ora_validate(b, m, m->player_id);                                   /* :25-27 */
  if (err) return err;
  int from_idx = board_idx(b, m->from_x, m->from_y), to_idx = board_idx(b, m->to_x, m->to_y);
  ora_tile* from = &b->t[from_idx];
  ora_tile* to = &b->t[to_idx];
  int32_t orig_owner = to->owner;                                                   /* :36-37 */
  int64_t orig_army = to->army;
  int64_t n;
  if (m->move_all) n = from->army - 1;                                              /* :41-42 */
  else { n = from->army / 2; if (n == 0) n = 1; }                                   /* :44-48 */
  from->army -= n;                                                                  /* :54 */
  if (changed_set) { tset_add(changed_set, from_idx); tset_add(changed_set, to_idx); } /* :57-60 */
  if (changed_bytes) { changed_bytes[from_idx] = 1; changed_bytes[to_idx] = 1; }
  if (to->owner == m->player_id) { to->army += n; return ORA_OK; }                  /* :62-66 */
  if (n > to->army) {                                                               /* :69-82 */
    to->owner = m->player_id;
    to->army = n - to->army;
    cap->x = m->to_x; cap->y = m->to_y; cap->tile_type = to->type;
    cap->capturing_player = m->player_id; cap->previous_owner = orig_owner; cap->previous_army = orig_army;
    *captured = 1;
  } else {
    to->army -= n;                                                                  /* :85 */
  }
  return ORA_OK;
}

int32_t ora_apply_move(ora_board* b, const ora_move* m, uint8_t* changed, ora_capture* cap, int32_t* captured) {
  ora_capture tmp; int32_t c = 0;
  int32_t err = apply_move(b, m, NULL, changed, cap ? cap : &tmp, &c);
  if (captured) *captured = c;
  return err;
}

/* core/movement.go:100-118  ProcessCaptures */
int32_t ora_process_captures(const ora_capture* caps, int32_t n, ora_elimination* out) {
  int32_t n_out = 0;
  for (int i = 0; i < n; i++) {
    const ora_capture* c = &caps[i];
    if (c->tile_type != ORA_TILE_GENERAL) continue;
    if (c->previous_owner == ORA_NEUTRAL) continue;
    if (c->previous_owner == c->capturing_player) continue;
    int seen = 0;                                              /* eliminationProcessedFor, :102,108 */
    for (int k = 0; k < n_out; k++) if (out[k].eliminated == c->previous_owner) seen = 1;
    if (seen) continue;
    out[n_out].eliminated = c->previous_owner;
    out[n_out].new_owner = c->capturing_player;
    n_out++;
  }
  return n_out;
}

/* ------------------------------------------------------------------------- */
/* game package: state.go:7-34, engine.go:17-43                              */
/* ------------------------------------------------------------------------- */
typedef struct ora_player {
  int32_t id, alive;
  int64_t army_count;
  int32_t general_idx;
  ilist   owned;     /* OwnedTiles */
} ora_player;

struct ora_engine {
  /* GameState */
  int32_t turn;
  ora_board* board;
  ora_player* players;
  int32_t num_players;
  tset changed;        /* ChangedTiles */
  tset vis_changed;    /* VisibilityChangedTiles */
  int32_t fog;         /* FogOfWarEnabled */
  /* Engine */
  int32_t game_over;
  int32_t original_players; /* WinConditionChecker.originalPlayers */
  ora_params params;
};

void ora_params_default(ora_params* p) {
  p->fog_of_war = 1;                 /* engine_initializer.go:118 */
  p->prod_general = 1;               /* internal/config/config.go:206 */
  p->prod_city = 1;                  /* :207 */
  p->prod_normal = 1;                /* :208 */
  p->normal_growth_interval = 25;    /* :209 */
}

ora_engine* ora_engine_new(int32_t w, int32_t h, int32_t players, const ora_params* params,
                           const int32_t* army, const int8_t* owner, const uint8_t* type) {
  ora_engine* e = (ora_engine*)calloc(1, sizeof(ora_engine));
  e->board = ora_board_new(w, h);
  int n = w * h;
  for (int i = 0; i < n; i++) {
    if (army) e->board->t[i].army = army[i];
    if (owner) e->board->t[i].owner = owner[i];
    if (type) e->board->t[i].type = type[i];
  }
  e->turn = 0;                                              /* engine_initializer.go:113-122 */
  e->num_players = players;
  e->players = (ora_player*)calloc((size_t)(players > 0 ? players : 1), sizeof(ora_player));
  tset_init(&e->changed, n);
  tset_init(&e->vis_changed, n);
  if (params) e->params = *params; else ora_params_default(&e->params);
  e->fog = e->params.fog_of_war;
  for (int i = 0; i < players; i++) {                       /* initializePlayers, :125-143 */
    int gidx = -1;
    for (int t = 0; t < n; t++)                             /* findPlayerGeneral :146-153 */
      if (e->board->t[t].type == ORA_TILE_GENERAL && e->board->t[t].owner == i) { gidx = t; break; }
    int64_t ac = 1;
    if (gidx >= 0) ac = e->board->t[gidx].army;
    e->players[i].id = i; e->players[i].alive = 1; e->players[i].general_idx = gidx; e->players[i].army_count = ac;
  }
  e->game_over = 0;
  e->original_players = players;                            /* :193 */
  return e;
}

void ora_engine_free(ora_engine* e) {
  if (!e) return;
  for (int i = 0; i < e->num_players; i++) free(e->players[i].owned.v);
  free(e->players); free(e->changed.bits); free(e->vis_changed.bits);
  ora_board_free(e->board); free(e);
}

/* ---- stats.go ------------------------------------------------------------ */
static void update_alive(ora_engine* e) {                   /* stats.go:52-61 / :133-142 */
  for (int p = 0; p < e->num_players; p++) e->players[p].alive = e->players[p].general_idx != -1;
}

static void perform_full_stats_update(ora_engine* e) {      /* stats.go:33-63 */
  int n = e->board->w * e->board->h;
  for (int p = 0; p < e->num_players; p++) {
    e->players[p].army_count = 0; e->players[p].general_idx = -1; e->players[p].owned.n = 0;
  }
  for (int idx = 0; idx < n; idx++) {
    const ora_tile* t = &e->board->t[idx];
    if (t->owner >= 0 && t->owner < e->num_players) {
      ora_player* p = &e->players[t->owner];
      p->army_count += t->army;
      ilist_push(&p->owned, idx);
      if (t->type == ORA_TILE_GENERAL) p->general_idx = idx;
    }
  }
  update_alive(e);
}

static void perform_incremental_stats_update(ora_engine* e) { /* stats.go:66-144 */
  int n = e->board->w * e->board->h;
  /* tempTileOwnership[tileIdx] = current owner, for tileIdx in ChangedTiles (:74-77) is
   * read back below straight from the board: the board is not written in between. */
  for (int pid = 0; pid < e->num_players; pid++) {          /* :90-130 */
    ora_player* pl = &e->players[pid];
    pl->army_count = 0; pl->general_idx = -1;
    int newn = 0;
    int oldn = pl->owned.n;
    for (int i = 0; i < oldn; i++) {                        /* :96-105 */
      int t = pl->owned.v[i];
      if (e->board->t[t].owner == pid) {
        pl->army_count += e->board->t[t].army;
        pl->owned.v[newn++] = t;
        if (e->board->t[t].type == ORA_TILE_GENERAL) pl->general_idx = t;
      }
    }
    pl->owned.n = newn;
    for (int t = 0; t < n; t++) {                           /* :108-127, ascending instead of map order */
      if (!e->changed.bits[t]) continue;
      if (e->board->t[t].owner != pid) continue;
      int found = 0;
      for (int k = 0; k < pl->owned.n; k++) if (pl->owned.v[k] == t) { found = 1; break; }
      if (!found) {
        pl->army_count += e->board->t[t].army;
        ilist_push(&pl->owned, t);
        if (e->board->t[t].type == ORA_TILE_GENERAL) pl->general_idx = t;
      }
    }
  }
  update_alive(e);
}

void ora_engine_update_player_stats(ora_engine* e) {        /* stats.go:8-30 */
  if (e->changed.count == 0 && e->turn > 0) return;         /* :10-14 */
  int threshold = (e->board->w * e->board->h) / 5;          /* :20 */
  if (e->turn == 0 || e->changed.count > threshold) { perform_full_stats_update(e); return; } /* :21-25 */
  perform_incremental_stats_update(e);                      /* :29 */
}

/* ---- visibility_optimized.go --------------------------------------------- */
static const int VIS_DX[9] = {-1, 0, 1, -1, 0, 1, -1, 0, 1}; /* :9-13 */
static const int VIS_DY[9] = {-1, -1, -1, 0, 0, 0, 1, 1, 1};

static void set_visibility_around(ora_engine* e, int tile_idx, uint32_t bit) { /* :119-129 */
  int x = tile_idx % e->board->w, y = tile_idx / e->board->w;
  for (int k = 0; k < 9; k++) {
    int nx = x + VIS_DX[k], ny = y + VIS_DY[k];
    if (in_bounds(e->board, nx, ny)) e->board->t[board_idx(e->board, nx, ny)].visible |= bit;
  }
}

static void clear_visibility_around(ora_engine* e, int tile_idx) { /* :132-150 */
  int x = tile_idx % e->board->w, y = tile_idx / e->board->w;
  uint32_t all = 0;
  for (int p = 0; p < e->num_players; p++) all |= (1u << (unsigned)p);
  uint32_t clear_mask = ~all;
  for (int k = 0; k < 9; k++) {
    int nx = x + VIS_DX[k], ny = y + VIS_DY[k];
    if (in_bounds(e->board, nx, ny)) e->board->t[board_idx(e->board, nx, ny)].visible &= clear_mask;
  }
}

static void perform_full_visibility_update(ora_engine* e) { /* :33-53 */
  int n = e->board->w * e->board->h;
  for (int i = 0; i < n; i++) e->board->t[i].visible = 0;
  for (int pid = 0; pid < e->num_players; pid++) {
    if (!e->players[pid].alive) continue;
    uint32_t bit = 1u << (unsigned)pid;
    for (int k = 0; k < e->players[pid].owned.n; k++) set_visibility_around(e, e->players[pid].owned.v[k], bit);
  }
}

static void perform_incremental_visibility_update(ora_engine* e) { /* :56-97 */
  int n = e->board->w * e->board->h;
  uint8_t affected[32]; memset(affected, 0, sizeof affected); /* tempAffectedPlayers */
  for (int t = 0; t < n; t++) {                             /* :68-73, collectAffectedPlayersOptimized :100-116 */
    if (!e->vis_changed.bits[t]) continue;
    int x = t % e->board->w, y = t / e->board->w;
    for (int dx = -2; dx <= 2; dx++)
      for (int dy = -2; dy <= 2; dy++) {
        int nx = x + dx, ny = y + dy;
        if (in_bounds(e->board, nx, ny)) {
          int owner = e->board->t[board_idx(e->board, nx, ny)].owner;
          if (owner >= 0 && owner < e->num_players) affected[owner] = 1;
        }
      }
  }
  for (int t = 0; t < n; t++) if (e->vis_changed.bits[t]) clear_visibility_around(e, t); /* :76-81 */
  for (int pid = 0; pid < e->num_players; pid++) {          /* :85-94 */
    if (!affected[pid]) continue;
    if (!e->players[pid].alive) continue;
    uint32_t bit = 1u << (unsigned)pid;
    for (int k = 0; k < e->players[pid].owned.n; k++) set_visibility_around(e, e->players[pid].owned.v[k], bit);
  }
}

void ora_engine_update_fog(ora_engine* e) {                 /* visibility.go:11-16 -> visibility_optimized.go:16-30 */
  if (!e->fog) return;                                      /* :17-19 */
  int threshold = (e->board->w * e->board->h) / 10;         /* :22 */
  if (e->turn == 0 || e->vis_changed.count > threshold) { perform_full_visibility_update(e); return; }
  perform_incremental_visibility_update(e);
}

/* ---- visibility.go:19-144: the LEGACY twin of the fog update (features.use_optimized_visibility = false) ----
 * Restated separately, in its own idiom (Tile.SetVisible per player and tile, no bit masks, its own loop
 * nests), as a second opinion on the optimized restatement above: the two must leave identical
 * VisibleBitfields after every update - same thresholds (visibility.go:24-25), same "affected = board owners
 * within 5x5" (:99-117), same re-lighting from OwnedTiles (:85-93).  They differ in ONE observable only:
 * this path goes through Tile.SetVisible and therefore accumulates DiscoveredBitfield (core/board.go:57-60),
 * the optimized path writes VisibleBitfield directly and leaves it 0.  Test-only. */
static void legacy_set_visibility_around(ora_engine* e, int tile_idx, int pid) {   /* visibility.go:119-131 */
  int x = tile_idx % e->board->w, y = tile_idx / e->board->w;
  for (int dx = -1; dx <= 1; dx++)
    for (int dy = -1; dy <= 1; dy++) {
      int nx = x + dx, ny = y + dy;
      if (nx >= 0 && nx < e->board->w && ny >= 0 && ny < e->board->h)
        ora_tile_set_visible(&e->board->t[ny * e->board->w + nx], pid, 1);
    }
}
static void legacy_clear_visibility_around(ora_engine* e, int tile_idx) {          /* visibility.go:133-146 */
  int x = tile_idx % e->board->w, y = tile_idx / e->board->w;
  for (int dx = -1; dx <= 1; dx++)
    for (int dy = -1; dy <= 1; dy++) {
      int nx = x + dx, ny = y + dy;
      if (nx >= 0 && nx < e->board->w && ny >= 0 && ny < e->board->h)
        for (int pid = 0; pid < e->num_players; pid++) ora_tile_set_visible(&e->board->t[ny * e->board->w + nx], pid, 0);
    }
}
void ora_engine_update_fog_legacy(ora_engine* e) {
  if (!e->fog) return;                                                              /* :19-21 */
  int n = e->board->w * e->board->h;
  if (e->turn == 0 || e->vis_changed.count > n / 10) {                              /* :24-28 performFullVisibilityUpdate :34-54 */
    for (int i = 0; i < n; i++)
      for (int pid = 0; pid < e->num_players; pid++) ora_tile_set_visible(&e->board->t[i], pid, 0);
    for (int pid = 0; pid < e->num_players; pid++) {
      if (!e->players[pid].alive) continue;
      for (int k = 0; k < e->players[pid].owned.n; k++) legacy_set_visibility_around(e, e->players[pid].owned.v[k], pid);
    }
    return;
  }
  /* performIncrementalVisibilityUpdate :57-95 */
  int affected[32]; for (int i = 0; i < 32; i++) affected[i] = 0;
  for (int t = 0; t < n; t++) {                                                     /* collectAffectedPlayers :99-117 */
    if (!e->vis_changed.bits[t]) continue;
    int x = t % e->board->w, y = t / e->board->w;
    for (int dx = -2; dx <= 2; dx++)
      for (int dy = -2; dy <= 2; dy++) {
        int nx = x + dx, ny = y + dy;
        if (nx >= 0 && nx < e->board->w && ny >= 0 && ny < e->board->h) {
          int owner = e->board->t[ny * e->board->w + nx].owner;
          if (owner >= 0 && owner < e->num_players) affected[owner] = 1;
        }
      }
  }
  for (int t = 0; t < n; t++) if (e->vis_changed.bits[t]) legacy_clear_visibility_around(e, t);   /* :75-81 */
  for (int pid = 0; pid < e->num_players; pid++) {                                  /* :85-93 */
    if (!affected[pid] || !e->players[pid].alive) continue;
    for (int k = 0; k < e->players[pid].owned.n; k++) legacy_set_visibility_around(e, e->players[pid].owned.v[k], pid);
  }
}
void ora_engine_player_visibility(const ora_engine* e, int32_t player, uint8_t* visible, uint8_t* fog) { /* :166-195 */
  int n = e->board->w * e->board->h;
  if (visible) memset(visible, 0, (size_t)n);
  if (fog) memset(fog, 0, (size_t)n);
  if (!e->fog) { if (visible) memset(visible, 1, (size_t)n); return; } /* :174-179 */
  uint32_t bit = (player >= 0 && player < 32) ? (1u << (unsigned)player) : 0; /* :182 (Go shifts; ids are < 32) */
  for (int i = 0; i < n; i++) {
    int v = (e->board->t[i].visible & bit) != 0;
    if (visible) visible[i] = (uint8_t)v;
    if (fog && !v && e->board->t[i].type != ORA_TILE_NORMAL) fog[i] = 1; /* :189-191 */
  }
}

/* ---- production_manager.go ------------------------------------------------ */
static int process_tile_production(const ora_engine* e, ora_tile* t, int grow_normal) { /* :76-101 */
  switch (t->type) {
    case ORA_TILE_GENERAL: t->army += e->params.prod_general; return e->params.prod_general;
    case ORA_TILE_CITY: t->army += e->params.prod_city; return e->params.prod_city;
    case ORA_TILE_NORMAL:
      if (grow_normal) { t->army += e->params.prod_normal; return e->params.prod_normal; }
      return 0;
    default: return 0;
  }
}

void ora_engine_process_production(ora_engine* e) {         /* :26-73, called with gs.Turn (engine.go:155-157) */
  int grow_normal = (e->turn % e->params.normal_growth_interval) == 0; /* :27 */
  for (int pid = 0; pid < e->num_players; pid++) {
    if (!e->players[pid].alive) continue;                   /* :40-42 */
    for (int k = 0; k < e->players[pid].owned.n; k++) {     /* :45-62 */
      int t = e->players[pid].owned.v[k];
      int prod = process_tile_production(e, &e->board->t[t], grow_normal);
      if (prod > 0) tset_add(&e->changed, t);               /* :59-61 */
    }
  }
}

/* ---- rules/win_conditions.go:21-57 ---------------------------------------- */
static void check_game_over_raw(const ora_engine* e, int* game_over, int* winner) {
  int alive = 0, last = 0;
  for (int p = 0; p < e->num_players; p++) if (e->players[p].alive) { alive++; last = e->players[p].id; }
  int over = (e->original_players > 1) ? (alive <= 1) : (alive == 0); /* :40-44 */
  *game_over = over;
  *winner = (over && alive == 1) ? last : -1;               /* :46-52 */
}
void ora_engine_check_game_over(ora_engine* e) {            /* engine.go:160-194 */
  int over, winner; check_game_over_raw(e, &over, &winner); e->game_over = over;
}
int32_t ora_engine_is_game_over(const ora_engine* e) { return e->game_over; } /* engine.go:198 */
int32_t ora_engine_winner(const ora_engine* e) {            /* engine.go:248-263 */
  if (!e->game_over) return -1;
  int over, winner; check_game_over_raw(e, &over, &winner); return winner;
}

/* ---- engine_initializer.go:218-225 ---------------------------------------- */
void ora_engine_initial_setup(ora_engine* e) {
  ora_engine_update_player_stats(e);
  ora_engine_update_fog(e);
  ora_engine_check_game_over(e);
}

/* ---- engine.go:118-152 ----------------------------------------------------- */
static void handle_eliminations(ora_engine* e, const ora_elimination* orders, int n) {
  for (int k = 0; k < n; k++) {
    ora_player* victim = &e->players[orders[k].eliminated];
    for (int i = 0; i < victim->owned.n; i++) {             /* :130-137 */
      int t = victim->owned.v[i];
      if (e->board->t[t].owner == orders[k].eliminated) {
        e->board->t[t].owner = orders[k].new_owner;
        tset_add(&e->changed, t);
        tset_add(&e->vis_changed, t);
      }
    }
    victim->alive = 0;                                      /* :140-141 */
    victim->general_idx = -1;
  }
}

/* ---- processor/action_processor.go:36-99 + engine.go:80-115 ---------------- */
static int32_t process_actions(ora_engine* e, const ora_move* actions_in, int n) {
  ora_move* actions = (ora_move*)malloc(sizeof(ora_move) * (size_t)(n > 0 ? n : 1));
  if (n > 0) memcpy(actions, actions_in, sizeof(ora_move) * (size_t)n);
  /* sort.Slice by PlayerID (:39-41).  Go's pdqsort uses insertion sort for n <= 12,
   * which is stable; the same insertion sort is used here for every n. */
  for (int i = 1; i < n; i++)
    for (int j = i; j > 0 && actions[j].player_id < actions[j - 1].player_id; j--) {
      ora_move tmp = actions[j]; actions[j] = actions[j - 1]; actions[j - 1] = tmp;
    }
  int32_t first_err = 0;
  ora_capture* caps = (ora_capture*)malloc(sizeof(ora_capture) * (size_t)(n > 0 ? n : 1));
  int ncap = 0;
  for (int i = 0; i < n; i++) {
    int pid = actions[i].player_id;
    if (pid < 0 || pid >= e->num_players || !e->players[pid].alive) continue; /* :56-60 */
    ora_capture cap; int32_t captured = 0;
    int32_t err = apply_move(e->board, &actions[i], &e->changed, NULL, &cap, &captured); /* :65 */
    if (err) { if (!first_err) first_err = err; continue; } /* :66-77 */
    if (captured) {                                         /* :78-87 */
      caps[ncap++] = cap;
      tset_add(&e->vis_changed, board_idx(e->board, cap.x, cap.y)); /* merged at engine.go:96-98 */
    }
  }
  if (ncap > 0) {                                           /* engine.go:101-109 */
    ora_elimination* orders = (ora_elimination*)malloc(sizeof(ora_elimination) * (size_t)ncap);
    int no = ora_process_captures(caps, ncap, orders);
    if (no > 0) { handle_eliminations(e, orders, no); ora_engine_update_player_stats(e); }
    free(orders);
  }
  free(caps); free(actions);
  return first_err;                                         /* engine.go:111-114 */
}

/* ---- turn_processor.go:29-77 ----------------------------------------------- */
int32_t ora_engine_step(ora_engine* e, const ora_move* actions, int32_t n) {
  /* validateGameState (:95-113).  A finished engine has also left PhaseRunning
   * (engine.go:178-185) and would fail the phase check first with an unwrapped
   * error; both cases are reported as ErrGameOver here. */
  if (e->game_over) return ORA_ERR_GAME_OVER;
  e->turn++;                                                /* initializeTurn :124-135 */
  ora_engine_update_fog(e);
  tset_clear(&e->changed);
  tset_clear(&e->vis_changed);
  int32_t err = process_actions(e, actions, n);             /* :55-57 */
  if (err) return err;
  ora_engine_process_production(e);                         /* :60 */
  ora_engine_update_player_stats(e);                        /* :65,170-179 */
  ora_engine_check_game_over(e);
  return ORA_OK;
}

/* ---- rules/legal_moves.go:19-73 via engine.go:271-280 ----------------------- */
void ora_engine_legal_mask(const ora_engine* e, int32_t player, uint8_t* mask) {
  int w = e->board->w, h = e->board->h;
  memset(mask, 0, (size_t)(w * h * 4));
  if (player < 0 || player >= e->num_players) return;       /* engine.go:273-276 */
  const ora_player* pl = &e->players[player];
  if (!pl->alive) return;                                   /* legal_moves.go:26-28 */
  static const int dx[4] = {0, 1, 0, -1}, dy[4] = {-1, 0, 1, 0}; /* :33-34 */
  for (int k = 0; k < pl->owned.n; k++) {
    int t = pl->owned.v[k];
    const ora_tile* tile = &e->board->t[t];
    if (tile->owner != player || tile->army <= 1) continue; /* :41-43 */
    int x = t % w, y = t / w;
    for (int d = 0; d < 4; d++) {
      ora_move m = {player, x, y, x + dx[d], y + dy[d], 0};
      if (ora_validate(e->board, &m, player) == ORA_OK) mask[(y * w + x) * 4 + d] = 1; /* :64-68 */
    }
  }
}

/* ---- raw accessors ---------------------------------------------------------- */
ora_board* ora_engine_board(ora_engine* e) { return e->board; }
int32_t ora_engine_turn(const ora_engine* e) { return e->turn; }
void ora_engine_set_turn(ora_engine* e, int32_t t) { e->turn = t; }
void ora_engine_set_game_over(ora_engine* e, int32_t v) { e->game_over = v; }
void ora_engine_set_fog(ora_engine* e, int32_t enabled) { e->fog = enabled; }
int32_t ora_engine_num_players(const ora_engine* e) { return e->num_players; }
int32_t ora_player_alive(const ora_engine* e, int32_t p) { return e->players[p].alive; }
void ora_player_set_alive(ora_engine* e, int32_t p, int32_t a) { e->players[p].alive = a; }
int64_t ora_player_army_count(const ora_engine* e, int32_t p) { return e->players[p].army_count; }
int32_t ora_player_general_idx(const ora_engine* e, int32_t p) { return e->players[p].general_idx; }
void ora_player_set_general_idx(ora_engine* e, int32_t p, int32_t idx) { e->players[p].general_idx = idx; }
int32_t ora_player_num_owned(const ora_engine* e, int32_t p) { return e->players[p].owned.n; }
const int32_t* ora_player_owned(const ora_engine* e, int32_t p) { return e->players[p].owned.v; }
void ora_player_set_owned(ora_engine* e, int32_t p, const int32_t* tiles, int32_t n) {
  e->players[p].owned.n = 0;
  for (int i = 0; i < n; i++) ilist_push(&e->players[p].owned, tiles[i]);
}
int32_t ora_engine_changed_count(const ora_engine* e) { return e->changed.count; }
int32_t ora_engine_vis_changed_count(const ora_engine* e) { return e->vis_changed.count; }
const uint8_t* ora_engine_changed(const ora_engine* e) { return e->changed.bits; }
const uint8_t* ora_engine_vis_changed(const ora_engine* e) { return e->vis_changed.bits; }

/* ------------------------------------------------------------------------- */
/* batch of engines                                                          */
/* ------------------------------------------------------------------------- */
struct ora_batch {
  ora_engine** prev;    /* experience snapshots (GameState.Clone before the step) */
  int32_t num_envs, max_w, max_h, max_p, stride, mask_bytes;
  uint32_t agent_noop, agent_half; /* gvec_set_agent_mix thresholds */
  ora_params params;
  ora_engine** env;
  int32_t* episode;     /* re-deal counter per env (auto-reset) */
  int32_t pool_size; uint64_t pool_seed; int32_t* pool_w; int32_t* pool_h; int32_t* pool_p;
};

ora_batch* ora_batch_new(int32_t num_envs, int32_t max_w, int32_t max_h, int32_t max_p, const ora_params* params) {
  ora_batch* b = (ora_batch*)calloc(1, sizeof(ora_batch));
  b->num_envs = num_envs; b->max_w = max_w; b->max_h = max_h; b->max_p = max_p;
  b->stride = max_w * max_h;
  b->mask_bytes = ora_mask_bytes(b->stride);
  b->agent_noop = 6554u; b->agent_half = 19661u;
  if (params) b->params = *params; else ora_params_default(&b->params);
  b->env = (ora_engine**)calloc((size_t)num_envs, sizeof(ora_engine*));
  b->episode = (int32_t*)calloc((size_t)num_envs, sizeof(int32_t));
  return b;
}
void ora_batch_free(ora_batch* b) {
  if (!b) return;
  for (int i = 0; i < b->num_envs; i++) { ora_engine_free(b->env[i]); if (b->prev) ora_engine_free(b->prev[i]); }
  free(b->prev);
  free(b->env); free(b->episode); free(b->pool_w); free(b->pool_h); free(b->pool_p); free(b);
}
ora_engine* ora_batch_engine(ora_batch* b, int32_t env) { return b->env[env]; }

int32_t ora_batch_reset(ora_batch* b, const int32_t* env_ids, int32_t n, const int32_t* army, const int8_t* owner,
                        const uint8_t* type, const int32_t* w, const int32_t* h, const int32_t* p) {
  for (int i = 0; i < n; i++) {
    int id = env_ids ? env_ids[i] : i;
    if (id < 0 || id >= b->num_envs) return -4;
    if (w[i] < 1 || w[i] > b->max_w || h[i] < 1 || h[i] > b->max_h || p[i] < 1 || p[i] > b->max_p) return -1;
    ora_engine_free(b->env[id]);
    size_t off = (size_t)i * (size_t)b->stride;
    b->env[id] = ora_engine_new(w[i], h[i], p[i], &b->params, army + off, owner + off, type + off);
    ora_engine_initial_setup(b->env[id]);
    b->episode[id] = 0;
  }
  return 0;
}

/* The packed mask of include/generals_vec.h: per player four direction bit-planes of mask_bytes/4
 * bytes; bit t (LSB first, little-endian dwords) of plane d = mask[(t)*4 + d] of the reference. */
int32_t ora_mask_bytes(int32_t stride) {
  static const int slots[6] = {1, 2, 4, 7, 10, 16}; /* 64-tile slots of the kernel variants (gvec_kernels.hip pick_variant) */
  int need = (stride + 63) / 64, n = 16;
  for (int i = 0; i < 6; i++) if (slots[i] >= need) { n = slots[i]; break; }
  int fd = (stride <= 32 * (2 * n - 1)) ? 2 * n - 1 : 2 * n; /* dwords per plane (gvec_api.hip) */
  return 16 * fd;
}
static void pack_bool_mask(const uint8_t* mask /* [n][4] */, int n, uint8_t* o, int mask_bytes) {
  int plane = mask_bytes / 4;
  for (int t = 0; t < n; t++)
    for (int d = 0; d < 4; d++)
      if (mask[t * 4 + d]) o[d * plane + (t >> 3)] |= (uint8_t)(1u << (t & 7));
}
static void pack_legal_bits(const ora_batch* b, const ora_engine* e, uint8_t* out /* [max_p][mask_bytes] */, uint8_t* scratch) {
  memset(out, 0, (size_t)b->max_p * (size_t)b->mask_bytes);
  for (int p = 0; p < e->num_players; p++) {
    ora_engine_legal_mask(e, p, scratch);
    pack_bool_mask(scratch, e->board->w * e->board->h, out + (size_t)p * (size_t)b->mask_bytes, b->mask_bytes);
  }
}

static int32_t step_env(ora_batch* b, int id, const ora_action8* acts) {
  ora_engine* e = b->env[id];
  ora_move mv[32]; int n = 0;
  for (int p = 0; p < e->num_players; p++) {
    const ora_action8* a = &acts[p];
    if (!(a->flags & 1u)) continue;                          /* nil action: converters.go:106-108,125-127 */
    mv[n].player_id = p; mv[n].from_x = a->from_x; mv[n].from_y = a->from_y; mv[n].to_x = a->to_x; mv[n].to_y = a->to_y;
    mv[n].move_all = (a->flags & 2u) ? 0 : 1;                /* MoveAll = !half, converters.go:122 */
    n++;
  }
  return ora_engine_step(e, mv, n);
}

static void redeal_env(ora_batch* b, int id);

int32_t ora_batch_step(ora_batch* b, const ora_action8* actions, int32_t* err, uint8_t* legal_bits, int32_t threads) {
  (void)threads;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) if (threads > 1) num_threads(threads > 1 ? threads : 1)
#endif
  for (int id = 0; id < b->num_envs; id++) {
    if (!b->env[id]) { if (err) err[id] = -1; continue; }
    int32_t rc;
    if (actions[(size_t)id * (size_t)b->max_p].flags & 4u) rc = 0;   /* GVEC_ACT_SKIP_ENV: sits this call out */
    else if (b->pool_size > 0 && (b->env[id]->game_over || (actions[(size_t)id * (size_t)b->max_p].flags & 8u))) {
      redeal_env(b, id); rc = 0;   /* finished, or GVEC_ACT_RESET_ENV: the caller ends the episode */
    }
    else rc = step_env(b, id, actions + (size_t)id * (size_t)b->max_p);
    if (err) err[id] = rc;
    if (legal_bits) {
      uint8_t* scratch = (uint8_t*)malloc((size_t)b->stride * 4);
      pack_legal_bits(b, b->env[id], legal_bits + (size_t)id * (size_t)b->max_p * (size_t)b->mask_bytes, scratch);
      free(scratch);
    }
  }
  return 0;
}

int32_t ora_batch_legal_mask(ora_batch* b, uint8_t* legal_bits, int32_t threads) {
  (void)threads;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) if (threads > 1) num_threads(threads > 1 ? threads : 1)
#endif
  for (int id = 0; id < b->num_envs; id++) {
    if (!b->env[id]) continue;
    uint8_t* scratch = (uint8_t*)malloc((size_t)b->stride * 4);
    pack_legal_bits(b, b->env[id], legal_bits + (size_t)id * (size_t)b->max_p * (size_t)b->mask_bytes, scratch);
    free(scratch);
  }
  return 0;
}

int32_t ora_batch_read_state(ora_batch* b, int32_t env_begin, int32_t n, const ora_state_view* v) {
  if (env_begin < 0 || n < 0 || env_begin + n > b->num_envs) return -4;
  for (int i = 0; i < n; i++) {
    ora_engine* e = b->env[env_begin + i];
    if (!e) return -1;
    size_t to = (size_t)i * (size_t)b->stride, po = (size_t)i * (size_t)b->max_p;
    int nt = e->board->w * e->board->h;
    if (v->army) memset(v->army + to, 0, sizeof(int32_t) * (size_t)b->stride);
    if (v->owner) memset(v->owner + to, 0xFF, (size_t)b->stride);
    if (v->type) memset(v->type + to, 0, (size_t)b->stride);
    if (v->visible) memset(v->visible + to, 0, (size_t)b->stride);
    if (v->listed) memset(v->listed + to, 0xFF, (size_t)b->stride);
    if (v->changed) memset(v->changed + to, 0, (size_t)b->stride);
    if (v->vis_changed) memset(v->vis_changed + to, 0, (size_t)b->stride);
    for (int t = 0; t < nt; t++) {
      const ora_tile* tl = &e->board->t[t];
      if (v->army) v->army[to + t] = (int32_t)tl->army;
      if (v->owner) v->owner[to + t] = (int8_t)tl->owner;
      if (v->type) v->type[to + t] = (uint8_t)tl->type;
      if (v->visible) v->visible[to + t] = (uint8_t)(tl->visible & 0xFFu);
      if (v->changed) v->changed[to + t] = e->changed.bits[t];
      if (v->vis_changed) v->vis_changed[to + t] = e->vis_changed.bits[t];
    }
    if (v->turn) v->turn[i] = e->turn;
    if (v->done) v->done[i] = (uint8_t)e->game_over;
    if (v->winner) v->winner[i] = (int8_t)ora_engine_winner(e);
    if (v->width) v->width[i] = e->board->w;
    if (v->height) v->height[i] = e->board->h;
    if (v->players) v->players[i] = e->num_players;
    for (int p = 0; p < b->max_p; p++) {
      int live = p < e->num_players;
      if (v->alive) v->alive[po + p] = live ? (uint8_t)e->players[p].alive : 0;
      if (v->army_count) v->army_count[po + p] = live ? (int32_t)e->players[p].army_count : 0;
      if (v->tile_count) v->tile_count[po + p] = live ? e->players[p].owned.n : 0;
      if (v->general_idx) v->general_idx[po + p] = live ? e->players[p].general_idx : -1;
      if (live && v->listed)
        for (int k = 0; k < e->players[p].owned.n; k++) v->listed[to + e->players[p].owned.v[k]] = (int8_t)p;
    }
  }
  return 0;
}

int32_t ora_batch_write_state(ora_batch* b, int32_t env_begin, int32_t n, const ora_state_view* v) {
  if (env_begin < 0 || n < 0 || env_begin + n > b->num_envs) return -4;
  for (int i = 0; i < n; i++) {
    ora_engine* e = b->env[env_begin + i];
    if (!e) return -1;
    size_t to = (size_t)i * (size_t)b->stride, po = (size_t)i * (size_t)b->max_p;
    int nt = e->board->w * e->board->h;
    for (int t = 0; t < nt; t++) {
      ora_tile* tl = &e->board->t[t];
      if (v->army) tl->army = v->army[to + t];
      if (v->owner) tl->owner = v->owner[to + t];
      if (v->type) tl->type = v->type[to + t];
      if (v->visible) tl->visible = v->visible[to + t];
    }
    if (v->changed) { tset_clear(&e->changed); for (int t = 0; t < nt; t++) if (v->changed[to + t]) tset_add(&e->changed, t); }
    if (v->vis_changed) { tset_clear(&e->vis_changed); for (int t = 0; t < nt; t++) if (v->vis_changed[to + t]) tset_add(&e->vis_changed, t); }
    if (v->turn) e->turn = v->turn[i];
    if (v->done) e->game_over = v->done[i];
    for (int p = 0; p < e->num_players; p++) {
      if (v->alive) e->players[p].alive = v->alive[po + p];
      if (v->army_count) e->players[p].army_count = v->army_count[po + p];
      if (v->general_idx) e->players[p].general_idx = v->general_idx[po + p];
      if (v->listed) {
        e->players[p].owned.n = 0;
        for (int t = 0; t < nt; t++) if (v->listed[to + t] == p) ilist_push(&e->players[p].owned, t);
      }
    }
  }
  return 0;
}

/* ------------------------------------------------------------------------- */
/* internal/experience: serializer.go, rewards.go                             */
/* ------------------------------------------------------------------------- */
ora_engine* ora_engine_clone(const ora_engine* e) { /* state.go:37-70 */
  int n = e->board->w * e->board->h;
  ora_engine* c = (ora_engine*)calloc(1, sizeof(ora_engine));
  *c = *e;
  c->board = ora_board_new(e->board->w, e->board->h);
  memcpy(c->board->t, e->board->t, sizeof(ora_tile) * (size_t)n);
  c->players = (ora_player*)calloc((size_t)(e->num_players > 0 ? e->num_players : 1), sizeof(ora_player));
  for (int p = 0; p < e->num_players; p++) {
    c->players[p] = e->players[p];
    c->players[p].owned.v = NULL; c->players[p].owned.n = 0; c->players[p].owned.cap = 0;
    for (int k = 0; k < e->players[p].owned.n; k++) ilist_push(&c->players[p].owned, e->players[p].owned.v[k]);
  }
  tset_init(&c->changed, n); tset_init(&c->vis_changed, n);
  for (int t = 0; t < n; t++) { if (e->changed.bits[t]) tset_add(&c->changed, t); if (e->vis_changed.bits[t]) tset_add(&c->vis_changed, t); }
  return c;
}

void ora_state_to_tensor(const ora_engine* e, int32_t player, float* out) { /* serializer.go:37-109 */
  int w = e->board->w, h = e->board->h, n = w * h;
  memset(out, 0, sizeof(float) * (size_t)(9 * n));
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      int t = y * w + x;
      const ora_tile* tile = &e->board->t[t];
      int visible = !e->fog || ora_tile_is_visible_to(tile, player);   /* :50 */
      if (visible) out[7 * n + t] = 1.0f;                               /* :53-55 */
      if (!visible) { out[8 * n + t] = 1.0f; continue; }                /* :58-65 */
      if (tile->type == ORA_TILE_MOUNTAIN) { out[6 * n + t] = 1.0f; continue; } /* :68-71 */
      if (tile->type == ORA_TILE_CITY || tile->type == ORA_TILE_GENERAL) out[5 * n + t] = 1.0f; /* :74-76 */
      if (tile->owner == player) {                                      /* :79-89 */
        if (tile->army > 0) { float v = (float)tile->army / 1000.0f; if (v > 1.0f) v = 1.0f; out[0 * n + t] = v; }
        out[2 * n + t] = 1.0f;
      } else if (tile->owner >= 0) {                                    /* :90-100 */
        if (tile->army > 0) { float v = (float)tile->army / 1000.0f; if (v > 1.0f) v = 1.0f; out[1 * n + t] = v; }
        out[3 * n + t] = 1.0f;
      } else {
        out[4 * n + t] = 1.0f;                                          /* :101-104 */
      }
    }
}

void ora_serializer_mask(const ora_engine* e, int32_t player, uint8_t* mask) { /* serializer.go:112-176 */
  int w = e->board->w, h = e->board->h;
  memset(mask, 0, (size_t)(w * h * 4));
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      const ora_tile* tile = &e->board->t[y * w + x];
      if (tile->owner != player || tile->army < 2) continue;            /* :128-130 */
      int base = (y * w + x) * 4;
      if (y > 0 && e->board->t[(y - 1) * w + x].type != ORA_TILE_MOUNTAIN) mask[base + 0] = 1;     /* up :134-141 */
      if (y < h - 1 && e->board->t[(y + 1) * w + x].type != ORA_TILE_MOUNTAIN) mask[base + 1] = 1; /* down :144-151 */
      if (x > 0 && e->board->t[y * w + x - 1].type != ORA_TILE_MOUNTAIN) mask[base + 2] = 1;       /* left :154-161 */
      if (x < w - 1 && e->board->t[y * w + x + 1].type != ORA_TILE_MOUNTAIN) mask[base + 3] = 1;   /* right :164-171 */
    }
}

float ora_army_advantage(const ora_engine* e, int32_t player) { /* rewards.go:153-175 */
  int n = e->board->w * e->board->h;
  int64_t pa = 0, ea = 0;
  for (int t = 0; t < n; t++) {
    const ora_tile* tile = &e->board->t[t];
    if (tile->owner == player) pa += tile->army; else if (tile->owner >= 0) ea += tile->army;
  }
  int64_t total = pa + ea;
  if (total == 0) return 0.0f;
  return (float)(pa - ea) / (float)total;
}

void ora_city_changes(const ora_engine* prev, const ora_engine* cur, int32_t player, int32_t* gained, int32_t* lost) { /* :110-129 */
  int n = cur->board->w * cur->board->h; *gained = 0; *lost = 0;
  for (int t = 0; t < n; t++) {
    if (cur->board->t[t].type != ORA_TILE_CITY) continue;
    int po = prev->board->t[t].owner, co = cur->board->t[t].owner;
    if (po != player && co == player) (*gained)++;
    if (po == player && co != player) (*lost)++;
  }
}

float ora_calculate_reward(const ora_engine* prev, const ora_engine* cur, int32_t player) { /* rewards.go:45-85, config :23-37 */
  const float WinGame = 1.0f, LoseGame = -1.0f, CaptureCity = 0.1f, LoseCity = -0.1f, CaptureGeneral = 0.5f, LoseGeneral = -0.5f,
              TerritoryGained = 0.01f, ArmyGained = 0.001f, ArmyAdvantage = 0.05f;
  volatile float reward = 0.0f; /* volatile: every += rounds to float32 like Go on amd64 (no fused multiply-add) */
  { /* GameState.IsGameOver / GetWinner (state.go:73-100) */
    int alive = 0, last = -1;
    for (int p = 0; p < cur->num_players; p++) if (cur->players[p].alive) { alive++; last = cur->players[p].id; }
    if (alive <= 1) {
      int winner = (alive == 1) ? last : -1;
      if (winner == player) return WinGame;
      else if (winner != -1) return LoseGame;
    }
  }
  int n = cur->board->w * cur->board->h;
  int64_t pt = 0, ct = 0, pa = 0, ca = 0;
  for (int t = 0; t < n; t++) {
    if (prev->board->t[t].owner == player) { pt++; pa += prev->board->t[t].army; }   /* :87-107 */
    if (cur->board->t[t].owner == player) { ct++; ca += cur->board->t[t].army; }
  }
  reward += (float)(ct - pt) * TerritoryGained;   /* :59-62 */
  reward += (float)(ca - pa) * ArmyGained;        /* :65-68 */
  int32_t cg, cl; ora_city_changes(prev, cur, player, &cg, &cl);
  reward += (float)cg * CaptureCity;              /* :71-73 */
  reward += (float)cl * LoseCity;
  int gg = 0, gl = 0;                             /* countGeneralChanges :132-151 */
  for (int t = 0; t < n; t++) {
    if (cur->board->t[t].type != ORA_TILE_GENERAL) continue;
    int po = prev->board->t[t].owner, co = cur->board->t[t].owner;
    if (po != player && po >= 0 && co == player) gg++;
    if (po == player && co != player) gl++;
  }
  reward += (float)gg * CaptureGeneral;           /* :76-78 */
  reward += (float)gl * LoseGeneral;
  reward += ora_army_advantage(cur, player) * ArmyAdvantage; /* :81-82 */
  return reward;
}

int32_t ora_batch_experience_begin(ora_batch* b) {
  if (!b->prev) b->prev = (ora_engine**)calloc((size_t)b->num_envs, sizeof(ora_engine*));
  for (int i = 0; i < b->num_envs; i++) { ora_engine_free(b->prev[i]); b->prev[i] = b->env[i] ? ora_engine_clone(b->env[i]) : NULL; }
  return 0;
}
int32_t ora_batch_rewards(ora_batch* b, float* rewards, uint8_t* done) {
  if (!b->prev) return -1;
  for (int i = 0; i < b->num_envs; i++) {
    const ora_engine* cur = b->env[i]; const ora_engine* prev = b->prev[i];
    int alive = 0;
    for (int p = 0; p < cur->num_players; p++) alive += cur->players[p].alive ? 1 : 0;
    if (done) done[i] = (uint8_t)(alive <= 1);   /* GameState.IsGameOver, state.go:73-82 */
    for (int p = 0; p < b->max_p; p++) {
      float r = 0.0f;
      /* a board re-dealt by auto-reset has no meaningful predecessor: reward 0 */
      if (p < cur->num_players && prev && prev->board->w == cur->board->w && prev->board->h == cur->board->h && cur->turn > prev->turn)
        r = ora_calculate_reward(prev, cur, p);
      rewards[(size_t)i * (size_t)b->max_p + p] = r;
    }
  }
  return 0;
}
int32_t ora_batch_observe(ora_batch* b, int32_t player, float* out) {
  for (int i = 0; i < b->num_envs; i++) {
    float* o = out + (size_t)i * 9 * (size_t)b->stride;
    memset(o, 0, sizeof(float) * 9 * (size_t)b->stride);
    ora_state_to_tensor(b->env[i], player, o);
  }
  return 0;
}
int32_t ora_batch_serializer_mask(ora_batch* b, uint8_t* bits) {
  uint8_t* scratch = (uint8_t*)malloc((size_t)b->stride * 4);
  memset(bits, 0, (size_t)b->num_envs * (size_t)b->max_p * (size_t)b->mask_bytes);
  for (int i = 0; i < b->num_envs; i++) {
    const ora_engine* e = b->env[i];
    for (int p = 0; p < e->num_players; p++) {
      ora_serializer_mask(e, p, scratch);
      pack_bool_mask(scratch, e->board->w * e->board->h, bits + ((size_t)i * (size_t)b->max_p + (size_t)p) * (size_t)b->mask_bytes,
                     b->mask_bytes);
    }
  }
  free(scratch);
  return 0;
}

/* ------------------------------------------------------------------------- */
/* synthetic inputs: the build's own spec (DESIGN.md "Synthetic inputs")      */
/* ------------------------------------------------------------------------- */
uint32_t ora_fmix32(uint32_t h) {
  h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16; return h;
}
static inline uint32_t mulhi32(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * (uint64_t)b) >> 32); }
static uint32_t env_key(uint64_t seed, uint32_t env) {
  uint32_t lo = (uint32_t)seed, hi = (uint32_t)(seed >> 32);
  return ora_fmix32(ora_fmix32(lo ^ 0x9E3779B9u) + hi * 0x85EBCA77u + env * 0xC2B2AE3Du + 0x27D4EB2Fu);
}

/* the agent's mixer: two rounds of xorshift + 24-bit multiply (low 32 bits of a 24x24-bit product) */
static inline uint32_t mul24(uint32_t a, uint32_t b) { return (uint32_t)((uint64_t)(a & 0xFFFFFFu) * (uint64_t)(b & 0xFFFFFFu)); }
uint32_t ora_amix(uint32_t x) {
  x ^= x >> 15; x = mul24(x, 0xE8A54Du); x ^= x >> 13; x = mul24(x, 0xAA34A7u); x ^= x >> 15; return x;
}

/* DESIGN.md "Synthetic inputs": per alive player two hashes h1, h2; no action if (h1 & 0xFFFF) < noop; half move if
 * (h1 >> 16) < half; with invalid_permille an unchecked move; else the kk-th legal move of
 * Engine.GetLegalActionMask(p), kk = ((h2 >> 16) * count) >> 16, in the order (t >> 5, d, t & 31): 32-tile blocks
 * ascending, inside a block direction by direction (up, right, down, left), inside a direction tiles ascending. */
static void agent_env(const ora_batch* b, const ora_engine* e, uint32_t ek, int32_t invalid_permille,
                      ora_action8* out, uint8_t* scratch) {
  static const int dx[4] = {0, 1, 0, -1}, dy[4] = {-1, 0, 1, 0};
  int w = e->board->w, h = e->board->h, n = w * h, n4 = n * 4;
  for (int p = 0; p < b->max_p; p++) memset(&out[p], 0, sizeof(ora_action8));
  for (int p = 0; p < e->num_players; p++) {
    if (!e->players[p].alive) continue;
    uint32_t h1 = ora_amix(mul24((uint32_t)p, 0x4A7C15u) + ek + (uint32_t)e->turn * 0x9E3779B1u + 0x165667B1u);
    if ((h1 & 0xFFFFu) < b->agent_noop) continue;             /* no-op, default p ~ 0.1 */
    int half = (h1 >> 16) < b->agent_half;                   /* default p ~ 0.3 */
    uint32_t h2 = ora_amix(h1 ^ 0x68E31DA4u);
    uint32_t hi16 = h2 >> 16;
    int t, d;
    if (invalid_permille > 0 && (mul24(h2 & 0xFFFFu, 1000u) >> 16) < (uint32_t)invalid_permille) {
      t = (int)(mul24(hi16, (uint32_t)n) >> 16);             /* unchecked move (H5 stress) */
      d = (int)((h1 >> 8) & 3u);
    } else {
      ora_engine_legal_mask(e, p, scratch);
      int cnt = 0;
      for (int i = 0; i < n4; i++) cnt += scratch[i];
      if (cnt == 0) continue;
      int k = (int)(mul24(hi16, (uint32_t)cnt) >> 16);
      t = -1; d = 0;
      for (int blk = 0; blk * 32 < n && t < 0; blk++)
        for (int dd = 0; dd < 4 && t < 0; dd++)
          for (int tt = blk * 32; tt < blk * 32 + 32 && tt < n; tt++)
            if (scratch[tt * 4 + dd]) { if (k == 0) { t = tt; d = dd; break; } k--; }
    }
    int x = t % w, y = t / w;
    out[p].from_x = (int8_t)x; out[p].from_y = (int8_t)y;
    out[p].to_x = (int8_t)(x + dx[d]); out[p].to_y = (int8_t)(y + dy[d]);
    out[p].flags = (uint8_t)(1u | (half ? 2u : 0u));
  }
}

int32_t ora_batch_set_agent_mix(ora_batch* b, int32_t noop_per_65536, int32_t half_per_65536) {
  b->agent_noop = (uint32_t)noop_per_65536; b->agent_half = (uint32_t)half_per_65536;
  return 0;
}
int32_t ora_batch_agent_actions(ora_batch* b, uint64_t seed, int32_t invalid_permille, ora_action8* actions, int32_t threads) {
  (void)threads;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) if (threads > 1) num_threads(threads > 1 ? threads : 1)
#endif
  for (int id = 0; id < b->num_envs; id++) {
    ora_action8* out = actions + (size_t)id * (size_t)b->max_p;
    if (!b->env[id] || b->env[id]->game_over) { memset(out, 0, sizeof(ora_action8) * (size_t)b->max_p); continue; }
    uint8_t* scratch = (uint8_t*)malloc((size_t)b->stride * 4);
    agent_env(b, b->env[id], env_key(seed, (uint32_t)id), invalid_permille, out, scratch);
    free(scratch);
  }
  return 0;
}

/* ---- map generator: mapgen/generator.go:64-253 over an RNG interface ------------------------------------------------
 * Two RNGs drive it: the build's counter RNG (ora_mapgen: what the device's parallel generator mirrors bit for bit) and
 * Go's own math/rand (ora_mapgen_go: seed -> the board game.NewEngine starts from), below. */
typedef struct map_rng {
  void* st;
  int (*intn)(void* st, int n);   /* rng.Intn(n) */
  int (*shuf)(void* st, int n);   /* the j of rng.Shuffle's swap(i, j), n = i + 1 */
} map_rng;

/* cfg: veins, min_len, max_len, city_ratio, city_army, spacing, stages (1 mountains | 2 cities | 4 generals) */
static int32_t mapgen_with(map_rng* R, int w, int h, int players, const int32_t* cfg, int32_t* army, int8_t* owner, uint8_t* type) {
  int n = w * h;
  int veins = cfg[0], min_len = cfg[1], max_len = cfg[2], city_ratio = cfg[3], city_army = cfg[4], spacing = cfg[5], stages = cfg[6];
  for (int i = 0; i < n; i++) { army[i] = 0; owner[i] = -1; type[i] = ORA_TILE_NORMAL; }
  /* placeMountains :77-142 */
  for (int v = 0; v < veins && (stages & 1); v++) {
    int sx = -1, sy = -1, found = 0;
    for (int a = 0; a < 100; a++) {
      int x = R->intn(R->st, w), y = R->intn(R->st, h);
      int idx = y * w + x;
      if (type[idx] == ORA_TILE_NORMAL && owner[idx] == -1) { sx = x; sy = y; found = 1; break; }
    }
    if (!found) continue;
    int cx = sx, cy = sy;
    type[cy * w + cx] = ORA_TILE_MOUNTAIN; army[cy * w + cx] = 0;
    int len = min_len;
    if (max_len > min_len) len += R->intn(R->st, max_len - min_len + 1);
    for (int i = 1; i < len; i++) {
      int dx[4] = {0, 1, 0, -1}, dy[4] = {-1, 0, 1, 0};
      for (int a = 3; a > 0; a--) {                          /* rand.Shuffle: Fisher-Yates from the top */
        int j = R->shuf(R->st, a + 1);
        int t = dx[a]; dx[a] = dx[j]; dx[j] = t; t = dy[a]; dy[a] = dy[j]; dy[j] = t;
      }
      int cand_x[4], cand_y[4], nc = 0;
      for (int j = 0; j < 4; j++) {
        int nx = cx + dx[j], ny = cy + dy[j];
        if (nx >= 0 && nx < w && ny >= 0 && ny < h) {
          int ni = ny * w + nx;
          if (type[ni] == ORA_TILE_NORMAL && owner[ni] == -1) { cand_x[nc] = nx; cand_y[nc] = ny; nc++; }
        }
      }
      if (nc == 0) break;
      int pick = R->intn(R->st, nc);
      cx = cand_x[pick]; cy = cand_y[pick];
      type[cy * w + cx] = ORA_TILE_MOUNTAIN; army[cy * w + cx] = 0;
    }
  }
  /* placeCities :144-164 */
  if (stages & 2) {
    int want = n / city_ratio, placed = 0, attempts = 0, max_attempts = want * 20;
    while (placed < want && attempts < max_attempts) {
      int x = R->intn(R->st, w), y = R->intn(R->st, h);
      int idx = y * w + x;
      if (owner[idx] == -1 && type[idx] == ORA_TILE_NORMAL) { type[idx] = ORA_TILE_CITY; army[idx] = city_army; placed++; }
      attempts++;
    }
  }
  /* placeGenerals :166-253 */
  int gx[32], gy[32];
  for (int pid = 0; pid < players && (stages & 4); pid++) {
    int placed_idx = -1;
    for (int a = 0; a < n; a++) {
      int x = R->intn(R->st, w), y = R->intn(R->st, h);
      int idx = y * w + x;
      if (owner[idx] != -1 || type[idx] != ORA_TILE_NORMAL) continue;
      int ok = 1;
      for (int o = 0; o < pid; o++) {
        int dxx = x - gx[o]; if (dxx < 0) dxx = -dxx;
        int dyy = y - gy[o]; if (dyy < 0) dyy = -dyy;
        if (dxx + dyy < spacing) { ok = 0; break; }
      }
      if (ok) { placed_idx = idx; break; }
    }
    if (placed_idx < 0) {                                    /* fallback scan :223-250 */
      for (int idx = 0; idx < n; idx++) {
        if (owner[idx] != -1 || type[idx] != ORA_TILE_NORMAL) continue;
        int x = idx % w, y = idx / w, ok = 1;
        for (int o = 0; o < pid; o++) {
          int dxx = x - gx[o]; if (dxx < 0) dxx = -dxx;
          int dyy = y - gy[o]; if (dyy < 0) dyy = -dyy;
          if (dxx + dyy < spacing) { ok = 0; break; }
        }
        if (ok) { placed_idx = idx; break; }
      }
    }
    if (placed_idx < 0) return -1;                           /* :252 */
    owner[placed_idx] = (int8_t)pid; army[placed_idx] = 2; type[placed_idx] = ORA_TILE_GENERAL; /* :175-178 */
    gx[pid] = placed_idx % w; gy[pid] = placed_idx / w;
  }
  return 0;
}

/* DefaultMapConfig, generator.go:25-47 with config.go:198-200 defaults */
static void default_map_cfg(int w, int h, int32_t* cfg) {
  int spacing = 5; if (spacing > w / 2 + h / 2) spacing = w / 2 + h / 2;
  cfg[0] = (w * h) / 50; cfg[1] = 3; cfg[2] = w / 4; cfg[3] = 20; cfg[4] = 40; cfg[5] = spacing; cfg[6] = 7;
}

/* the build's counter RNG */
typedef struct mrng { uint32_t key, ctr; } mrng;
static inline uint32_t mdraw(mrng* r) { return ora_fmix32(r->key + (r->ctr++) * 0x9E3779B9u); }
static int mintn(void* r, int n) { return (int)mulhi32(mdraw((mrng*)r), (uint32_t)n); }

int32_t ora_mapgen(uint64_t seed, int32_t env, int32_t w, int32_t h, int32_t players,
                   int32_t* army, int8_t* owner, uint8_t* type) {
  mrng r; r.key = ora_fmix32(env_key(seed, (uint32_t)env) ^ 0x5BD1E995u); r.ctr = 0;
  map_rng R = {&r, mintn, mintn};
  int32_t cfg[7];
  default_map_cfg(w, h, cfg);
  return mapgen_with(&R, w, h, players, cfg, army, owner, type);
}

/* ---- Go's math/rand (go.mod:3: go 1.24.0), restated from its published algorithm ----------------------------------
 * rand.New(rand.NewSource(seed)): an additive lagged Fibonacci generator x[n] = x[n-607] + x[n-273] mod 2^64 (rng.go:
 * rngLen 607, rngTap 273), seeded by the LCG x = 48271 x mod (2^31 - 1) - three draws per word, shifted 40 / 20 / 0 -
 * XORed with the 607-word table rngCooked.  The table is derived, not copied: scripts/gen_go_rand_cooked.py.
 * Int63 = low 63 bits; Int31 = Int63 >> 32; Int31n(n): mask for powers of two, else rejection above
 * 2^31 - 1 - 2^31 % n, then % n; Intn(n) = Int31n for n < 2^31; Shuffle's j = int31n(i + 1): Lemire's multiply-shift on
 * Uint32 = Int63 >> 31 with rejection below (-n) % n.
 * Pinned by: Seed(1) -> Intn(100) = 81 87 47 59 81 18 25 40 56 0 (Go's well-known default sequence) and, through
 * ora_mapgen_go, the reference's seed-12345 vectors (mapgen/generator_test.go:61-85, :396-455). */
static const uint64_t go_rng_cooked[607] = {
#include "go_rand_cooked.inc"
};
struct ora_gorand { uint64_t vec[607]; int tap, feed; };

static int32_t go_seedrand(int32_t x) {
  int32_t hi = x / 44488, lo = x % 44488;
  x = 48271 * lo - 3399 * hi;
  if (x < 0) x += 2147483647;
  return x;
}
void ora_gorand_seed(ora_gorand* r, int64_t seed) {
  r->tap = 0; r->feed = 607 - 273;
  seed %= 2147483647;
  if (seed < 0) seed += 2147483647;
  if (seed == 0) seed = 89482311;
  int32_t x = (int32_t)seed;
  for (int i = -20; i < 607; i++) {
    x = go_seedrand(x);
    if (i >= 0) {
      uint64_t u = (uint64_t)x << 40;
      x = go_seedrand(x); u ^= (uint64_t)x << 20;
      x = go_seedrand(x); u ^= (uint64_t)x;
      r->vec[i] = u ^ go_rng_cooked[i];
    }
  }
}
int64_t ora_gorand_int63(ora_gorand* r) {
  if (--r->tap < 0) r->tap += 607;
  if (--r->feed < 0) r->feed += 607;
  uint64_t x = r->vec[r->feed] + r->vec[r->tap];
  r->vec[r->feed] = x;
  return (int64_t)(x & 0x7FFFFFFFFFFFFFFFull);
}
int32_t ora_gorand_intn(ora_gorand* r, int32_t n) {            /* Intn -> Int31n */
  if ((n & (n - 1)) == 0) return (int32_t)(ora_gorand_int63(r) >> 32) & (n - 1);
  int32_t max = (int32_t)(2147483647u - (2147483648u % (uint32_t)n));
  int32_t v = (int32_t)(ora_gorand_int63(r) >> 32);
  while (v > max) v = (int32_t)(ora_gorand_int63(r) >> 32);
  return v % n;
}
static int32_t go_int31n(ora_gorand* r, int32_t n) {            /* rand.go int31n (Shuffle, Perm) */
  uint32_t v = (uint32_t)(ora_gorand_int63(r) >> 31);
  uint64_t prod = (uint64_t)v * (uint64_t)(uint32_t)n;
  uint32_t low = (uint32_t)prod;
  if (low < (uint32_t)n) {
    uint32_t thresh = (uint32_t)(-n) % (uint32_t)n;
    while (low < thresh) {
      v = (uint32_t)(ora_gorand_int63(r) >> 31);
      prod = (uint64_t)v * (uint64_t)(uint32_t)n;
      low = (uint32_t)prod;
    }
  }
  return (int32_t)(prod >> 32);
}
ora_gorand* ora_gorand_new(int64_t seed) {
  ora_gorand* r = (ora_gorand*)malloc(sizeof(ora_gorand));
  ora_gorand_seed(r, seed);
  return r;
}
void ora_gorand_free(ora_gorand* r) { free(r); }
static int go_intn_cb(void* r, int n) { return ora_gorand_intn((ora_gorand*)r, n); }
static int go_shuf_cb(void* r, int n) { return go_int31n((ora_gorand*)r, n); }

/* mapgen.NewGenerator(cfg, rand.New(rand.NewSource(seed))).GenerateMap() (generator.go:56-75).  cfg7 NULL:
 * DefaultMapConfig(w, h, players) - then this is the board game.NewEngine(ctx, GameConfig{Width, Height, Players,
 * Rng: rand.New(rand.NewSource(seed))}) starts from (engine_initializer.go:106-110) - else {veins, min_len, max_len,
 * city_ratio, city_army, spacing, stages} as the reference's tests override them. */
int32_t ora_mapgen_go(int64_t seed, int32_t w, int32_t h, int32_t players, const int32_t* cfg7,
                      int32_t* army, int8_t* owner, uint8_t* type) {
  ora_gorand* r = ora_gorand_new(seed);
  map_rng R = {r, go_intn_cb, go_shuf_cb};
  int32_t cfg[7];
  if (cfg7) memcpy(cfg, cfg7, sizeof cfg); else default_map_cfg(w, h, cfg);
  int32_t rc = mapgen_with(&R, w, h, players, cfg, army, owner, type);
  ora_gorand_free(r);
  return rc;
}

/* auto-reset pool: board j = ora_mapgen(pool_seed, j, pool_w[j], pool_h[j], pool_p[j]) */
int32_t ora_batch_set_pool(ora_batch* b, int32_t pool_size, uint64_t seed, const int32_t* w, const int32_t* h, const int32_t* p) {
  free(b->pool_w); free(b->pool_h); free(b->pool_p);
  b->pool_size = pool_size; b->pool_seed = seed;
  b->pool_w = (int32_t*)malloc(sizeof(int32_t) * (size_t)pool_size);
  b->pool_h = (int32_t*)malloc(sizeof(int32_t) * (size_t)pool_size);
  b->pool_p = (int32_t*)malloc(sizeof(int32_t) * (size_t)pool_size);
  for (int j = 0; j < pool_size; j++) {
    b->pool_w[j] = w ? w[j] : b->max_w; b->pool_h[j] = h ? h[j] : b->max_h; b->pool_p[j] = p ? p[j] : b->max_p;
  }
  return 0;
}

static void redeal_env(ora_batch* b, int id) {
  b->episode[id]++;
  uint32_t hk = ora_fmix32(env_key(b->pool_seed, (uint32_t)id) ^ ((uint32_t)b->episode[id] * 0x9E3779B1u));
  int j = (int)mulhi32(hk, (uint32_t)b->pool_size);
  int w = b->pool_w[j], h = b->pool_h[j], p = b->pool_p[j];
  int32_t* army = (int32_t*)malloc(sizeof(int32_t) * (size_t)(w * h));
  int8_t* owner = (int8_t*)malloc((size_t)(w * h));
  uint8_t* type = (uint8_t*)malloc((size_t)(w * h));
  ora_mapgen(b->pool_seed, j, w, h, p, army, owner, type);
  ora_engine_free(b->env[id]);
  b->env[id] = ora_engine_new(w, h, p, &b->params, army, owner, type);
  ora_engine_initial_setup(b->env[id]);
  free(army); free(owner); free(type);
}

int64_t ora_batch_rollout(ora_batch* b, int32_t turns, uint64_t seed, int32_t invalid_permille, int32_t threads) {
  return ora_batch_rollout_masks(b, turns, seed, invalid_permille, threads, NULL);
}

/* The same rollout; legal_bits != NULL: after every turn an env also packs its players' legal masks into its slot of
 * legal_bits [B][max_p][mask_bytes], like gvec_step / per-turn gvec_rollout do on the device - the work the timed CPU
 * baseline must include to be the same workload as the GPU leg (bench.py). */
int64_t ora_batch_rollout_masks(ora_batch* b, int32_t turns, uint64_t seed, int32_t invalid_permille, int32_t threads, uint8_t* legal_bits) {
  int64_t steps = 0;
  (void)threads;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) reduction(+ : steps) if (threads > 1) num_threads(threads > 1 ? threads : 1)
#endif
  for (int id = 0; id < b->num_envs; id++) {
    if (!b->env[id]) continue;
    uint8_t* scratch = (uint8_t*)malloc((size_t)b->stride * 4);
    ora_action8 acts[32];
    uint32_t ek = env_key(seed, (uint32_t)id);
    for (int k = 0; k < turns; k++) {
      if (b->env[id]->game_over) {
        if (b->pool_size > 0) { redeal_env(b, id); continue; }
        break;
      }
      agent_env(b, b->env[id], ek, invalid_permille, acts, scratch);
      step_env(b, id, acts);
      if (legal_bits) pack_legal_bits(b, b->env[id], legal_bits + (size_t)id * (size_t)b->max_p * (size_t)b->mask_bytes, scratch);
      steps++;
    }
    free(scratch);
  }
  return steps;
}

/* For every env of a batch: what the LEGACY update would make of the state the next Step will start from
 * (Turn + 1, visibility.go:11-32 runs inside initializeTurn, turn_processor.go:124-135).  visible[B][stride]
 * (bit p), discovered[B][stride]; the batch itself is not touched. */
int32_t ora_batch_next_fog_legacy(ora_batch* b, uint8_t* visible, uint8_t* discovered) {
  for (int id = 0; id < b->num_envs; id++) {
    ora_engine* c = ora_engine_clone(b->env[id]);
    int n = c->board->w * c->board->h;
    c->turn += 1;
    ora_engine_update_fog_legacy(c);
    memset(visible + (size_t)id * (size_t)b->stride, 0, (size_t)b->stride);
    memset(discovered + (size_t)id * (size_t)b->stride, 0, (size_t)b->stride);
    for (int t = 0; t < n; t++) {
      visible[(size_t)id * (size_t)b->stride + (size_t)t] = (uint8_t)(c->board->t[t].visible & 0xFFu);
      discovered[(size_t)id * (size_t)b->stride + (size_t)t] = (uint8_t)(c->board->t[t].discovered & 0xFFu);
    }
    ora_engine_free(c);
  }
  return 0;
}

