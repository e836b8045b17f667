"""Batched experience collection with the reference's record layout.

Mirrors SimpleCollector.OnStateTransition (internal/experience/collector.go:30-98) for B engines:
for every (env, player) that submitted an action this turn it yields the (s, a, r, s', done, mask)
tuple the reference streams as experiencepb.Experience, and `as_dicts` emits exactly the dict
keys python/experience_stream_client.py:134-158 (`_process_experience`) hands to trainers.

All tensors come from the device kernels of the experience side channel (gvec_observe,
gvec_serializer_mask, gvec_experience_begin / _rewards): nothing is recomputed on the host.

Reference quirk, not copied: on the gRPC path MoveAction.From/To are never set
(internal/grpc/gameserver/converters.go:116-123), so turn_processor.go:195-199 makes the reference
record action index 0 for every experience; here `action` is the real Serializer.ActionToIndex.
"""
import numpy as np

from .vec_engine import ACT_VALID, unpack_legal_bits


def action_to_index(actions, width):
    """Serializer.ActionToIndex (internal/experience/serializer.go:179-198): (y*W+x)*4 + dir with
    dir 0 up, 1 down, 2 left, 3 right (0 when the move is not a unit step, like the Go code)."""
    fx, fy = actions["from_x"].astype(np.int64), actions["from_y"].astype(np.int64)
    dx = actions["to_x"].astype(np.int64) - fx
    dy = actions["to_y"].astype(np.int64) - fy
    d = np.zeros(fx.shape, np.int64)
    d = np.where((dy == 1) & (dx == 0), 1, d)
    d = np.where((dy == 0) & (dx == -1), 2, d)
    d = np.where((dy == 0) & (dx == 1), 3, d)
    return (fy * width + fx) * 4 + d


def index_to_action(index, width, height):
    """Serializer.IndexToAction (internal/experience/serializer.go:201-223): -> (from_x, from_y, to_x, to_y);
    like the Go code, no bounds check on the destination."""
    index = np.asarray(index, np.int64)
    d, t = index % 4, index // 4
    fx, fy = t % width, t // width
    tx = fx + np.where(d == 3, 1, 0) - np.where(d == 2, 1, 0)
    ty = fy + np.where(d == 1, 1, 0) - np.where(d == 0, 1, 0)
    return fx, fy, tx, ty


class VecExperienceCollector:
    """Usage per turn:  c.before_step(); err = engine.step(actions); batch = c.after_step(actions)"""

    def __init__(self, engine, game_id_prefix="vec"):
        self.e = engine
        self.prefix = game_id_prefix
        self._state = self._mask = None
        self._serial = 0

    def before_step(self):
        """TurnProcessor.captureStateForExperience (turn_processor.go:116-121)."""
        self._state = self.e.observe(-1)                 # StateToTensor(prevState, player)   collector.go:43
        self._mask = self.e.serializer_mask_bits()       # GenerateActionMask(prevState, ...) collector.go:50
        self.e.experience_begin()

    def after_step(self, actions):
        """-> dict over the K (env, player) pairs that acted (collector.go:33-37).  Every experience is shaped by
        ITS env's board (TensorState.shape = [9, Board.H, Board.W], collector.go:64,71): in a padded batch of
        mixed sizes `state` / `next_state` / `action_mask` are lists of per-experience arrays; when every env has
        the same size they are stacked arrays, as before."""
        e = self.e
        acted = (np.asarray(actions["flags"]) & ACT_VALID) != 0           # [B, P]
        env, player = np.nonzero(acted)
        rewards, done = e.experience_rewards()                             # CalculateReward, IsGameOver  :47,56
        nxt = e.observe(-1)                                                # StateToTensor(currState, player) :44
        st = e.game_state(fields=("turn", "width", "height"))
        w, h = st["width"][env].astype(np.int64), st["height"][env].astype(np.int64)
        # observe / serializer_mask lay an env's tensor out with ITS OWN row pitch at the start of the padded slot
        # (include/generals_vec.h: index c*H*W + y*W + x; mask bit t = y*W + x of plane d)
        def tensors(src):
            return [src[ev, pl, : 9 * hh * ww].reshape(9, hh, ww) for ev, pl, ww, hh in zip(env, player, w, h)]
        state, next_state = tensors(self._state), tensors(nxt)
        masks = [unpack_legal_bits(self._mask[ev, pl], int(ww), int(hh)) for ev, pl, ww, hh in zip(env, player, w, h)]
        action = np.array([action_to_index(actions[ev, pl], int(ww)) for ev, pl, ww in zip(env, player, w)], np.int32).reshape(-1)
        if len(env) and (w == w[0]).all() and (h == h[0]).all():
            state, next_state, masks = np.stack(state), np.stack(next_state), np.stack(masks)
        return {
            "env": env, "player_id": player.astype(np.int32), "turn": st["turn"][env], "width": w, "height": h,
            "state": state, "next_state": next_state, "action": action,
            "reward": rewards[env, player], "done": done[env], "action_mask": masks,
        }

    def as_dicts(self, batch):
        """One dict per experience with the keys of ExperienceStreamClient._process_experience."""
        out = []
        for k in range(len(batch["env"])):
            self._serial += 1
            out.append({
                "experience_id": f"{self.prefix}-{self._serial}", "game_id": f"{self.prefix}-env{int(batch['env'][k])}",
                "player_id": int(batch["player_id"][k]), "turn": int(batch["turn"][k]), "state": batch["state"][k],
                "action": int(batch["action"][k]), "reward": float(batch["reward"][k]), "next_state": batch["next_state"][k],
                "done": bool(batch["done"][k]), "action_mask": batch["action_mask"][k],
            })
        return out
