// engine_kat.cpp — the reference's engine tests (internal/game/engine_test.go, action_mask_test.go) written
// against the C++ host mirror gvec::VecEngine (generalsreinforcementlearning_amd/host/vec_engine.hpp), i.e.
// through the C ABI on the GPU.  Built and run by tests/test_hip_cpp_mirror.py.  Boards are laid out by hand
// (the reference uses its seeded map generator); every expected value is the one the Go test asserts.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "generalsreinforcementlearning_amd/host/vec_engine.hpp"

namespace {
int g_fail = 0;
#define EXPECT(cond)                                                        \
  do {                                                                      \
    if (!(cond)) {                                                          \
      std::fprintf(stderr, "FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond);  \
      ++g_fail;                                                             \
    }                                                                       \
  } while (0)

enum : uint8_t { Normal = GVEC_TILE_NORMAL, General = GVEC_TILE_GENERAL, City = GVEC_TILE_CITY, Mountain = GVEC_TILE_MOUNTAIN };
struct Tile { int x, y, owner, army; uint8_t type; };

struct Board {
  int w, h, players;
  std::vector<int32_t> army;
  std::vector<int8_t> owner;
  std::vector<uint8_t> type;
  Board(int w_, int h_, int p_, const std::vector<Tile>& tiles) : w(w_), h(h_), players(p_), army(w_ * h_, 0), owner(w_ * h_, -1), type(w_ * h_, Normal) {
    for (const Tile& t : tiles) {
      const int i = t.y * w + t.x;  // core/board.go:108
      army[i] = t.army; owner[i] = static_cast<int8_t>(t.owner); type[i] = t.type;
    }
  }
  void reset(gvec::VecEngine& e) const { e.Reset(army, owner, type, {w}, {h}, {players}); }
};

gvec::GameConfig cfg(int w, int h, int p) {
  gvec::GameConfig c;
  c.NumEnvs = 1; c.Width = w; c.Height = h; c.Players = p;
  return c;
}

// engine_test.go:65-86 TestEngine_Step_BasicTurn
void step_basic_turn() {
  gvec::VecEngine e(cfg(5, 5, 1));
  Board(5, 5, 1, {{2, 2, 0, 2, General}, {4, 0, -1, 40, City}}).reset(e);
  gvec::GameState s0 = e.GetGameState(0, 1);
  EXPECT(s0.Turn[0] == 0 && !s0.GameOver[0]);
  const std::vector<int32_t>& err = e.Step({{}});  // no actions
  EXPECT(err[0] == 0);
  gvec::GameState s1 = e.GetGameState(0, 1);
  EXPECT(s1.Turn[0] == s0.Turn[0] + 1);                // :82
  EXPECT(s1.ArmyCount[0] == s0.ArmyCount[0] + 1);      // :84 general production
  EXPECT(!s1.GameOver[0]);                             // :85
}

// engine_test.go:88-102 TestEngine_Step_GameOverReturnError
void step_game_over_returns_error() {
  gvec::VecEngine e(cfg(5, 5, 1));
  Board(5, 5, 1, {{2, 2, 0, 2, General}}).reset(e);
  uint8_t done = 1;  // engine.gameOver = true (:98)
  gvec_state_view v{};
  v.done = &done;
  gvec::check(gvec_write_state(e.handle(), 0, 1, &v, GVEC_MEM_HOST), "gvec_write_state");
  EXPECT(e.Step({{}})[0] == GVEC_ERR_GAME_OVER);  // :101 core.ErrGameOver
}

// engine_test.go:104-182 TestEngine_ProcessTurnProduction: growth on turn 25, none on turn 24
void production_turn_25_vs_24() {
  for (int target : {25, 24}) {
    gvec::VecEngine e(cfg(5, 5, 1));
    Board(5, 5, 1, {{2, 2, 0, 2, General}, {0, 0, 0, 5, City}, {1, 0, 0, 2, Normal}}).reset(e);
    int32_t turn = target - 1;  // Step increments first (turn_processor.go:124-135)
    gvec_state_view v{};
    v.turn = &turn;
    gvec::check(gvec_write_state(e.handle(), 0, 1, &v, GVEC_MEM_HOST), "gvec_write_state");
    EXPECT(e.Step({{}})[0] == 0);
    gvec::GameState s = e.GetGameState(0, 1);
    EXPECT(s.Turn[0] == target);
    EXPECT(s.Army[2 * 5 + 2] == 3);                        // general +1 every turn (:159,:176)
    EXPECT(s.Army[0] == 6);                                // city +1 every turn (:160,:177)
    EXPECT(s.Army[1] == (target == 25 ? 3 : 2));           // normal tile only on the growth interval (:161 vs :178)
  }
}

// engine_test.go:184-248 TestEngine_PlayerEliminationAndTileTurnover
void elimination_and_tile_turnover() {
  gvec::VecEngine e(cfg(5, 5, 2));
  Board(5, 5, 2, {{4, 4, 0, 2, General}, {0, 0, 0, 20, Normal}, {0, 1, 1, 1, General}, {1, 1, 1, 5, City}, {2, 2, 1, 3, Normal}}).reset(e);
  EXPECT(e.Step({{{0, 0, 0, 0, 1, true}}})[0] == 0);  // P0 moves all from (0,0) onto P1's general at (0,1)
  gvec::GameState s = e.GetGameState(0, 1);
  EXPECT(s.Alive[0] == 1 && s.Alive[1] == 0);   // :228-229
  EXPECT(s.GeneralIdx[1] == -1);                // :230
  EXPECT(s.Owner[1 * 5 + 0] == 0 && s.Army[1 * 5 + 0] == 19);  // captured general: 19 - 1 + 1 production (:233-234)
  EXPECT(s.Owner[1 * 5 + 1] == 0 && s.Army[1 * 5 + 1] == 6);   // city handed over, then produces (:237-238)
  EXPECT(s.Owner[2 * 5 + 2] == 0 && s.Army[2 * 5 + 2] == 3);   // normal tile handed over (:241-242)
  EXPECT(s.GameOver[0] == 1 && s.Winner[0] == 0);              // :245-247
}

// engine_test.go:250-303 TestEngine_Step_ActionFromDeadPlayer
void action_from_dead_player_ignored() {
  gvec::VecEngine e(cfg(5, 5, 2));
  Board(5, 5, 2, {{4, 4, 0, 2, General}, {3, 0, 1, 2, General}}).reset(e);
  // the Go test then writes engine.gs directly (:263-277): P1's general tile becomes an empty neutral tile, P1 is
  // marked dead, P0 gets 10 armies on (0,0), P1 "formally owns" (1,1) with 5; OwnedTiles lists are left alone
  gvec::GameState s0 = e.GetGameState(0, 1);
  s0.Owner[3] = -1; s0.Army[3] = 0; s0.Type[3] = Normal;
  s0.Owner[0] = 0; s0.Army[0] = 10; s0.Type[0] = Normal;
  s0.Owner[1 * 5 + 1] = 1; s0.Army[1 * 5 + 1] = 5; s0.Type[1 * 5 + 1] = Normal;
  s0.Alive[1] = 0;
  s0.GeneralIdx[1] = -1;
  gvec_state_view v{};
  v.army = s0.Army.data(); v.owner = s0.Owner.data(); v.type = s0.Type.data();
  v.alive = s0.Alive.data(); v.general_idx = s0.GeneralIdx.data();
  gvec::check(gvec_write_state(e.handle(), 0, 1, &v, GVEC_MEM_HOST), "gvec_write_state");
  EXPECT(e.Step({{{1, 1, 1, 1, 2, true}, {0, 0, 0, 0, 1, true}}})[0] == 0);  // :287-288 no error
  gvec::GameState s = e.GetGameState(0, 1);
  EXPECT(s.Owner[1 * 5 + 1] == 1 && s.Army[1 * 5 + 1] == 5);   // the dead player's tile did not move (:291-292)
  EXPECT(s.Army[0] == 1);                                      // :295
  EXPECT(s.Owner[1 * 5 + 0] == 0 && s.Army[1 * 5 + 0] == 9);   // :298-299
  EXPECT(s.Alive[0] == 1 && s.Alive[1] == 0);                  // :301-302
}

// action_mask_test.go:57-100 "basic mask generation" and :170-204 "mask with mountains"
void legal_action_mask() {
  {
    gvec::VecEngine e(cfg(3, 3, 2));
    Board(3, 3, 2, {{1, 1, 0, 5, General}, {2, 2, 1, 1, General}}).reset(e);
    std::vector<bool> m = e.GetLegalActionMask(0, 0, 3, 3);
    EXPECT(m.size() == 36);  // :79
    int n = 0;
    for (bool b : m) n += b;
    EXPECT(n == 4);                                                  // :97
    for (int d = 0; d < 4; ++d) EXPECT(m[(1 * 3 + 1) * 4 + d]);     // :84-95 up, right, down, left from (1,1)
    EXPECT(e.GetLegalActionMask(0, -1, 3, 3) == std::vector<bool>(36, false));  // :226-236 invalid player
    EXPECT(e.GetLegalActionMask(0, 5, 3, 3) == std::vector<bool>(36, false));   // :238-243
  }
  {
    gvec::VecEngine e(cfg(3, 3, 1));
    Board(3, 3, 1, {{1, 1, 0, 5, General}, {1, 0, -1, 0, Mountain}, {2, 1, -1, 0, Mountain}}).reset(e);
    std::vector<bool> m = e.GetLegalActionMask(0, 0, 3, 3);
    const int base = (1 * 3 + 1) * 4;
    EXPECT(!m[base + 0]);  // up: mountain (:193)
    EXPECT(!m[base + 1]);  // right: mountain (:194)
    EXPECT(m[base + 2]);   // down (:195)
    EXPECT(m[base + 3]);   // left (:196)
  }
}
}  // namespace

// the same basic turn on a sharded engine (gvec_create_sharded: 5 envs over two shards, both on device 0 here)
void sharded_engine_basic_turn() {
  gvec::GameConfig c = cfg(5, 5, 1);
  c.NumEnvs = 5;
  c.Devices = {0, 0};
  gvec::VecEngine e(c);
  EXPECT(gvec_num_shards(e.Handle()) == 2);
  const Board b(5, 5, 1, {{2, 2, 0, 2, General}, {4, 0, -1, 40, City}});
  std::vector<int32_t> army, w(5, 5), h(5, 5), p(5, 1);
  std::vector<int8_t> owner;
  std::vector<uint8_t> type;
  for (int i = 0; i < 5; ++i) {
    army.insert(army.end(), b.army.begin(), b.army.end());
    owner.insert(owner.end(), b.owner.begin(), b.owner.end());
    type.insert(type.end(), b.type.begin(), b.type.end());
  }
  e.Reset(army, owner, type, w, h, p);
  const std::vector<int32_t>& err = e.Step({{}, {}, {}, {}, {}});
  gvec::GameState s = e.GetGameState(0, 5);
  for (int i = 0; i < 5; ++i) EXPECT(err[i] == 0 && s.Turn[i] == 1 && s.ArmyCount[i] == 3 && !s.GameOver[i]);
}

int main() {
  try {
    sharded_engine_basic_turn();
    step_basic_turn();
    step_game_over_returns_error();
    production_turn_25_vs_24();
    elimination_and_tile_turnover();
    action_from_dead_player_ignored();
    legal_action_mask();
  } catch (const std::exception& ex) {
    std::fprintf(stderr, "exception: %s\n", ex.what());
    return 2;
  }
  if (g_fail) {
    std::fprintf(stderr, "%d expectation(s) failed\n", g_fail);
    return 1;
  }
  std::puts("engine_kat: all expectations hold");
  return 0;
}
