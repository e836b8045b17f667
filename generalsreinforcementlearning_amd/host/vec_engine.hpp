// vec_engine.hpp — C++ host mirror of the reference's *game.Engine method set for B boards,
// over the C ABI of include/generals_vec.h (header-only; link with -lgvec_hip).
//
// Names follow internal/game/engine.go: NewEngine :62, Step :75, GameState :197, IsGameOver :198,
// GetWinner :248, GetLegalActionMask :271, GetChangedTiles :283, GetVisibilityChangedTiles :292,
// ComputePlayerVisibility (visibility.go:153).  Errors: API misuse throws gvec::Error; game-rule
// errors come back per env as the sentinel codes of core/errors.go:8-17.
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/generals_vec.h"

namespace gvec {

struct Error : std::runtime_error {
  int32_t code;
  Error(int32_t c, const std::string& what) : std::runtime_error(what + ": " + gvec_last_error()), code(c) {}
};
inline void check(int32_t rc, const char* what) {
  if (rc < 0) throw Error(rc, what);
}

// core.MoveAction (core/action.go:23-36)
struct MoveAction {
  int PlayerID, FromX, FromY, ToX, ToY;
  bool MoveAll;
};

// game.GameConfig (engine.go:45-58) for a batch
struct GameConfig {
  int NumEnvs = 1, Width = 20, Height = 20, Players = 2, Device = 0;
  bool FogOfWarEnabled = true;  // engine_initializer.go:118
  bool AutoReset = false;
  std::vector<int32_t> Devices;  // non-empty: one engine over several GPUs (gvec_create_sharded; Device is then ignored)
};

struct GameState {  // state.go:25-34 + Player (state.go:7-13), env-major planes
  std::vector<int32_t> Army, Turn, ArmyCount, GeneralIdx, TileCount;
  std::vector<int8_t> Owner, Winner, Listed;
  std::vector<uint8_t> Type, VisibleBitfield, ChangedTiles, VisibilityChangedTiles, GameOver, Alive;
};

class VecEngine {
 public:
  explicit VecEngine(const GameConfig& cfg) : cfg_(cfg) {  // NewEngine, engine.go:62-71
    gvec_config c;
    check(gvec_config_default(&c), "gvec_config_default");
    c.num_envs = cfg.NumEnvs;
    c.max_width = cfg.Width;
    c.max_height = cfg.Height;
    c.max_players = cfg.Players;
    c.device = cfg.Device;
    c.fog_of_war = cfg.FogOfWarEnabled ? 1 : 0;
    c.auto_reset = cfg.AutoReset ? 1 : 0;
    if (!cfg.Devices.empty()) check(gvec_create_sharded(&c, cfg.Devices.data(), static_cast<int32_t>(cfg.Devices.size()), &h_), "gvec_create_sharded");
    else check(gvec_create(&c, &h_), "gvec_create");
    stride_ = gvec_tile_stride(h_);
    mask_bytes_ = gvec_mask_bytes(h_);
    actions_.resize(static_cast<size_t>(cfg.NumEnvs) * cfg.Players);
    err_.resize(cfg.NumEnvs);
    mask_.resize(static_cast<size_t>(cfg.NumEnvs) * cfg.Players * mask_bytes_);
  }
  ~VecEngine() {
    if (h_) gvec_destroy(h_);
  }
  VecEngine(const VecEngine&) = delete;
  VecEngine& operator=(const VecEngine&) = delete;

  int TileStride() const { return stride_; }
  gvec_handle* Handle() const { return h_; }  // for the entry points this mirror does not wrap

  // boards: [n][stride] planes, tile index y*W+x (core/board.go:108); runs performInitialSetup
  void Reset(const std::vector<int32_t>& army, const std::vector<int8_t>& owner, const std::vector<uint8_t>& type,
             const std::vector<int32_t>& w, const std::vector<int32_t>& h, const std::vector<int32_t>& p) {
    check(gvec_reset(h_, nullptr, static_cast<int32_t>(w.size()), army.data(), owner.data(), type.data(), w.data(), h.data(),
                     p.data(), GVEC_MEM_HOST),
          "gvec_reset");
  }
  void ResetGenerated(uint64_t seed) { check(gvec_reset_generated(h_, seed, nullptr, nullptr, nullptr), "gvec_reset_generated"); }

  // Engine.Step (engine.go:75): actions[env] = that env's moves this turn. Returns per-env sentinel codes.
  const std::vector<int32_t>& Step(const std::vector<std::vector<MoveAction>>& actions) {
    for (auto& a : actions_) a = gvec_action{};
    for (size_t env = 0; env < actions.size(); ++env)
      for (const MoveAction& m : actions[env]) {
        if (m.PlayerID < 0 || m.PlayerID >= cfg_.Players) continue;  // action_processor.go:56-60
        gvec_action& g = actions_[env * cfg_.Players + m.PlayerID];
        g.from_x = clamp8(m.FromX);
        g.from_y = clamp8(m.FromY);
        g.to_x = clamp8(m.ToX);
        g.to_y = clamp8(m.ToY);
        g.flags = static_cast<uint8_t>(GVEC_ACT_VALID | (m.MoveAll ? 0u : GVEC_ACT_HALF));
      }
    check(gvec_step(h_, actions_.data(), err_.data(), mask_.data(), GVEC_MEM_HOST), "gvec_step");
    return err_;
  }

  // Engine.GetLegalActionMask(playerID) of one env (engine.go:271-280), from the last Step/reset
  std::vector<bool> GetLegalActionMask(int env, int playerID, int w, int h) {
    std::vector<bool> out(static_cast<size_t>(w) * h * 4, false);
    if (playerID < 0 || playerID >= cfg_.Players) return out;  // engine.go:273-276
    check(gvec_legal_mask(h_, mask_.data(), GVEC_MEM_HOST), "gvec_legal_mask");
    const uint8_t* bits = &mask_[(static_cast<size_t>(env) * cfg_.Players + playerID) * mask_bytes_];
    const size_t plane = static_cast<size_t>(mask_bytes_) / 4;  // four direction bit-planes (generals_vec.h, gvec_step)
    for (size_t t = 0; t < static_cast<size_t>(w) * h; ++t)
      for (size_t d = 0; d < 4; ++d) out[t * 4 + d] = (bits[d * plane + (t >> 3)] >> (t & 7)) & 1u;
    return out;
  }

  GameState GetGameState(int begin, int n) {  // GameState()/IsGameOver()/GetWinner()/GetChangedTiles()...
    GameState s;
    const size_t nt = static_cast<size_t>(n) * stride_, np = static_cast<size_t>(n) * cfg_.Players;
    s.Army.resize(nt); s.Owner.resize(nt); s.Type.resize(nt); s.VisibleBitfield.resize(nt); s.Listed.resize(nt);
    s.ChangedTiles.resize(nt); s.VisibilityChangedTiles.resize(nt);
    s.Turn.resize(n); s.GameOver.resize(n); s.Winner.resize(n);
    s.Alive.resize(np); s.ArmyCount.resize(np); s.GeneralIdx.resize(np); s.TileCount.resize(np);
    gvec_state_view v{};
    v.army = s.Army.data(); v.owner = s.Owner.data(); v.type = s.Type.data(); v.visible = s.VisibleBitfield.data();
    v.listed = s.Listed.data(); v.changed = s.ChangedTiles.data(); v.vis_changed = s.VisibilityChangedTiles.data();
    v.turn = s.Turn.data(); v.done = s.GameOver.data(); v.winner = s.Winner.data();
    v.alive = s.Alive.data(); v.army_count = s.ArmyCount.data(); v.general_idx = s.GeneralIdx.data(); v.tile_count = s.TileCount.data();
    check(gvec_read_state(h_, begin, n, &v, GVEC_MEM_HOST), "gvec_read_state");
    return s;
  }

  // Engine.ComputePlayerVisibility(playerID) (visibility.go:153)
  void ComputePlayerVisibility(int playerID, std::vector<uint8_t>* visible, std::vector<uint8_t>* fog) {
    visible->assign(static_cast<size_t>(cfg_.NumEnvs) * stride_, 0);
    fog->assign(static_cast<size_t>(cfg_.NumEnvs) * stride_, 0);
    check(gvec_player_visibility(h_, playerID, visible->data(), fog->data(), GVEC_MEM_HOST), "gvec_player_visibility");
  }

  gvec_rollout_stats Rollout(int turns, uint64_t seed, int invalid_permille = 0, bool fused = true) {
    gvec_rollout_stats st{};
    check(gvec_rollout(h_, turns, seed, invalid_permille, fused ? 1 : 0, &st), "gvec_rollout");
    return st;
  }

  gvec_handle* handle() { return h_; }

 private:
  static int8_t clamp8(int v) { return static_cast<int8_t>(v < -128 ? -128 : (v > 127 ? 127 : v)); }
  GameConfig cfg_;
  gvec_handle* h_ = nullptr;
  int stride_ = 0, mask_bytes_ = 0;
  std::vector<gvec_action> actions_;
  std::vector<int32_t> err_;
  std::vector<uint8_t> mask_;
};

}  // namespace gvec
