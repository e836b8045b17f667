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


# --------------------------------------------------------------------------------------------------
# experience records: the compact per-transition form a rank ships to the StreamAggregator side
# (include/generals_vec.h "experience records").  decode_records is the consumer: it expands a slab of
# records into the same batch VecExperienceCollector.after_step yields from the device tensors.
# --------------------------------------------------------------------------------------------------
def record_offsets(layout):
    """Dword offsets of the record fields for a layout dict (VecEngine.experience_record_layout)."""
    mp, fd, ns = layout["mp"], layout["fd"], layout["ns"]
    planes = 4 + 2 * mp
    mask = planes + (4 * mp + 3) * fd
    army_prev = mask + 4 * mp * fd
    return {"action": 4, "reward": 4 + mp, "planes": planes, "mask": mask, "army_prev": army_prev, "army_next": army_prev + ns * 32}


def _bits(words, n):
    """[..., fd] uint32 -> [..., n] bool (bit t of the string = bit t & 31 of dword t >> 5)."""
    b = np.unpackbits(np.ascontiguousarray(words).view(np.uint8), axis=-1, bitorder="little")
    return b[..., :n].astype(bool)


def state_to_tensor(owner_planes, vis_plane, gen, city, mtn, army, player, fog, w, h):
    """Serializer.StateToTensor (internal/experience/serializer.go:37-109) from bit-planes: owner_planes [P][N] bool,
    vis_plane / gen / city / mtn [N] bool, army [N] (any integer type; values >= 1000 all normalise to 1) -> float32 [9][h][w]."""
    n = w * h
    out = np.zeros((9, n), np.float32)
    visible = np.ones(n, bool) if not fog else vis_plane                                   # :50
    out[7] = visible                                                                        # :53-55
    out[8] = ~visible                                                                       # :58-65
    open_ = visible & ~mtn                                                                  # mountains short-circuit :68-71
    out[6] = visible & mtn
    out[5] = open_ & (gen | city)                                                           # :74-76
    mine = owner_planes[player]
    owned = owner_planes.any(0)
    norm = np.minimum(army.astype(np.float32) / np.float32(1000.0), np.float32(1.0))        # :82-85 float32 division, clamp
    arm = np.where(army > 0, norm, np.float32(0.0)).astype(np.float32)
    out[0] = np.where(open_ & mine, arm, 0.0)                                               # :79-89
    out[2] = open_ & mine
    out[1] = np.where(open_ & ~mine & owned, arm, 0.0)                                      # :90-100
    out[3] = open_ & ~mine & owned
    out[4] = open_ & ~owned                                                                 # :101-104
    return out.reshape(9, h, w)


def decode_records(slab, layout, drop_invalid=True):
    """slab: uint8 / uint32 array holding k records back to back (host memory).  Returns the batch dict of
    VecExperienceCollector.after_step (one entry per (record, player) that acted, record-major), plus
    "env" = the record's env id and "valid" (False: the env was re-dealt since the snapshot, its transition is void)."""
    rd = layout["record_dw"]
    recs = np.ascontiguousarray(slab).view(np.uint32).reshape(-1, rd)
    off, mp, fd = record_offsets(layout), layout["mp"], layout["fd"]
    out = {k: [] for k in ("env", "player_id", "turn", "width", "height", "state", "next_state", "action", "reward", "done",
                           "action_mask", "valid")}
    for r in recs:
        w, h, P, flags = int(r[1] & 0xFF), int((r[1] >> 8) & 0xFF), int((r[1] >> 16) & 0xFF), int(r[1] >> 24)
        n = w * h
        valid = bool(flags & 4)
        if drop_invalid and not valid:
            continue
        acted = int(r[2])
        if not acted:
            continue
        planes = _bits(r[off["planes"]: off["planes"] + (4 * mp + 3) * fd].reshape(4 * mp + 3, fd), n)
        prev_own, prev_vis, next_own, next_vis = planes[0:mp], planes[mp:2 * mp], planes[2 * mp:3 * mp], planes[3 * mp:4 * mp]
        gen, city, mtn = planes[4 * mp], planes[4 * mp + 1], planes[4 * mp + 2]
        masks = _bits(r[off["mask"]: off["mask"] + 4 * mp * fd].reshape(mp, 4, fd), n)              # [p][d][t]
        army_prev = r[off["army_prev"]: off["army_prev"] + layout["ns"] * 32].view(np.uint16)[:n]
        army_next = r[off["army_next"]: off["army_next"] + layout["ns"] * 32].view(np.uint16)[:n]
        actions = r[off["action"]: off["action"] + mp].view(np.int32)
        rewards = r[off["reward"]: off["reward"] + mp].view(np.float32)
        for p in range(P):
            if not (acted >> p) & 1:
                continue
            out["env"].append(int(r[3]))
            out["player_id"].append(p)
            out["turn"].append(int(r[0].view(np.int32)))
            out["width"].append(w)
            out["height"].append(h)
            out["state"].append(state_to_tensor(prev_own[:P], prev_vis[p], gen, city, mtn, army_prev, p, bool(flags & 2), w, h))
            out["next_state"].append(state_to_tensor(next_own[:P], next_vis[p], gen, city, mtn, army_next, p, bool(flags & 2), w, h))
            out["action"].append(int(actions[p]))
            out["reward"].append(rewards[p])
            out["done"].append(bool(flags & 1))
            out["action_mask"].append(np.moveaxis(masks[p], 0, 1).reshape(n * 4))                    # index t*4 + d
            out["valid"].append(valid)
    res = {"env": np.array(out["env"], np.int64), "player_id": np.array(out["player_id"], np.int32),
           "turn": np.array(out["turn"], np.int32), "width": np.array(out["width"], np.int64), "height": np.array(out["height"], np.int64),
           "action": np.array(out["action"], np.int32), "reward": np.array(out["reward"], np.float32),
           "done": np.array(out["done"], bool), "valid": np.array(out["valid"], bool),
           "state": out["state"], "next_state": out["next_state"], "action_mask": out["action_mask"]}
    if len(res["env"]) and (res["width"] == res["width"][0]).all() and (res["height"] == res["height"][0]).all():
        for k in ("state", "next_state", "action_mask"):
            res[k] = np.stack(res[k])
    return res


class RecordExpander:
    """decode_records on the GPU (gvec_expand_experience_records) with its output buffers kept: a consumer that expands a
    slab of k records every few steps allocates once.  `expand(records)` fills, for every (record, player) slot, record-major,
    state / next_state float32 [k * MP, 9 * stride] (an experience's own [9, H, W] at the start of its row), action_mask
    uint8 [k * MP, 4 * stride] and meta int32 [k * MP, 8] (present, env id, player, turn, action, reward bits, done,
    W | H << 8) and returns them as a dict of views; `compact(...)` turns that into decode_records' batch dict (torch
    tensors, one entry per slot that holds an experience).  The kernel runs at HBM speed (57.6 KB written per experience
    at 20x20); decode_records' per-record numpy loop is the readable twin the tests compare it with."""

    def __init__(self, layout, max_records, device):
        import ctypes as C

        import torch

        from . import _lib
        self._t, self._C, self._lib, self.L = torch, C, _lib, _lib.load()
        self.layout, self.max_records, self.device = dict(layout), int(max_records), torch.device(device)
        rd, mp, stride = layout["record_dw"], layout["mp"], layout["stride"]
        self._lay = (C.c_int32 * 8)(rd, mp, layout["fd"], layout["ns"], layout["max_players"], stride, 0, 0)
        n = self.max_records * mp
        self.state = torch.empty((n, 9 * stride), dtype=torch.float32, device=self.device)
        self.next_state = torch.empty_like(self.state)
        self.mask = torch.empty((n, 4 * stride), dtype=torch.uint8, device=self.device)
        self.meta = torch.empty((n, 8), dtype=torch.int32, device=self.device)

    def expand(self, records, stream=None):
        t, C = self._t, self._C
        flat = records.contiguous().view(t.uint8).reshape(-1)
        k = flat.numel() // (self.layout["record_dw"] * 4)
        if k > self.max_records or flat.device != self.device:
            raise ValueError(f"{k} records on {flat.device}: this expander holds {self.max_records} on {self.device}")
        s = t.cuda.current_stream(self.device).cuda_stream if stream is None else stream
        self._lib.check(self.L.gvec_expand_experience_records(self.device.index or 0, C.c_void_p(int(s)), self._lay, C.c_void_p(flat.data_ptr()), k,
                                                              C.c_void_p(self.state.data_ptr()), C.c_void_p(self.next_state.data_ptr()),
                                                              C.c_void_p(self.mask.data_ptr()), C.c_void_p(self.meta.data_ptr())),
                        "gvec_expand_experience_records")
        n = k * self.layout["mp"]
        return {"state": self.state[:n], "next_state": self.next_state[:n], "action_mask": self.mask[:n], "meta": self.meta[:n]}

    def compact(self, slots):
        t = self._t
        keep = t.nonzero(slots["meta"][:, 0] != 0).reshape(-1)
        m = slots["meta"][keep]
        return {"env": m[:, 1].to(t.int64), "player_id": m[:, 2], "turn": m[:, 3], "action": m[:, 4],
                "reward": m[:, 5].contiguous().view(t.float32), "done": m[:, 6] != 0, "width": (m[:, 7] & 0xFF).to(t.int64),
                "height": ((m[:, 7] >> 8) & 0xFF).to(t.int64), "state": slots["state"][keep], "next_state": slots["next_state"][keep],
                "action_mask": slots["action_mask"][keep].view(t.bool)}


def expand_records_device(records, layout, stream=None):
    """One-shot form: `records` a CUDA uint8 / int32 tensor holding k records back to back (what RecordGather /
    gvec_gather_experience_records delivered) -> decode_records' batch dict as torch tensors on that device."""
    k = records.numel() * records.element_size() // (layout["record_dw"] * 4)
    ex = RecordExpander(layout, max(1, k), records.device)
    return ex.compact(ex.expand(records, stream))


class RecordReplayRing:
    """A replay memory of COMPACT experience records in HBM: what `ExperienceDataset` (python/experience_stream_client.py:
    326-378: fill a buffer from the stream, draw random training batches) is to a learner, without the experiences ever
    being expanded until they are drawn.  A 20x20 4-player record is 3.7 KB and holds the transition of every player that
    acted; expanded it is 4 x 30 KB of tensors - so 288 GB hold ~70 million transitions' records where they would hold
    2.5 million expanded ones.  `append_step` has the engine write the records of a step straight into the ring (no staging
    copy); `sample(k)` draws k records uniformly without replacement, gathers them (k x 3.7 KB) and expands them on the GPU
    (gvec_expand_experience_records, HBM speed) into decode_records' batch dict - one entry per player that acted, so a draw
    of k records yields between k and k * players experiences.  Oldest records are overwritten (ring)."""

    def __init__(self, engine, capacity_records, seed=0):
        import torch
        self._t, self.engine = torch, engine
        self.layout = engine.experience_record_layout()
        self.record_bytes = engine.experience_record_bytes()
        self.capacity = int(capacity_records)
        if self.capacity <= 0:
            raise ValueError(f"capacity must be positive, got {capacity_records}")
        self.device = torch.device("cuda", engine.device)
        self.ring = torch.zeros((self.capacity, self.record_bytes), dtype=torch.uint8, device=self.device)
        self.cursor = self.size = self.total_appended = 0
        self._gen = torch.Generator(device=self.device)
        self._gen.manual_seed(int(seed))
        self._expander = None

    def append_step(self, actions=None, env_begin=0, n=None, env_id_base=0):
        """The records of envs [env_begin, env_begin + n) for the step just played (gvec_experience_begin[_range] before it),
        written by the engine into the ring at the cursor - in two pieces when the run wraps."""
        n = self.engine.B - env_begin if n is None else int(n)
        if n > self.capacity:
            raise ValueError(f"{n} records per step do not fit a ring of {self.capacity}")
        first = min(n, self.capacity - self.cursor)
        self.engine.experience_records(self.ring[self.cursor].data_ptr(), actions, env_begin, first, env_id_base)
        if first < n:
            self.engine.experience_records(self.ring[0].data_ptr(), actions, env_begin + first, n - first, env_id_base)
        self.cursor = (self.cursor + n) % self.capacity
        self.size = min(self.size + n, self.capacity)
        self.total_appended += n

    def __len__(self):
        return self.size

    def sample_indices(self, k):
        from ._sampling import distinct_indices
        return distinct_indices(self._t, self.size, k, self.device, self._gen)

    def sample(self, k, indices=None):
        """-> decode_records' batch dict as CUDA tensors (state / next_state [m, 9 * stride] with an experience's own
        [9, H, W] at the start of its row, action_mask [m, 4 * stride], env, player_id, turn, action, reward, done, width,
        height) for the m experiences the k drawn records hold.  The tensors are the expander's buffers: valid until the next
        sample()."""
        idx = self.sample_indices(k) if indices is None else indices
        if self._expander is None or self._expander.max_records < len(idx):
            self._expander = RecordExpander(self.layout, max(len(idx), 1), self.device)
        return self._expander.compact(self._expander.expand(self.ring[idx]))


class ExperienceBatcher:
    """BatchProcessor (internal/grpc/gameserver/batch_processor.go:12-166) as StreamAggregator configures it
    (stream_aggregator.go:64-69: 32 experiences or 100 ms, whichever comes first): add() returns the batches that
    became full, poll(now) flushes a non-empty partial batch whose timeout has passed, flush() forces it
    (Flush, :71-78).  Single-threaded and clock-injected: the caller owns the loop (the Go version owns goroutines)."""

    def __init__(self, batch_size=32, batch_timeout_s=0.1, clock=None):
        import time
        self.batch_size = 32 if batch_size <= 0 else int(batch_size)           # NewBatchProcessor :33-35
        self.batch_timeout_s = 0.1 if batch_timeout_s <= 0 else float(batch_timeout_s)   # :36-38
        self.clock = clock or time.monotonic
        self.current, self.last_flush = [], self.clock()

    def add(self, experiences):
        out = []
        for e in experiences:
            self.current.append(e)
            if len(self.current) >= self.batch_size:                             # :99-107: full -> flush, ticker reset
                out.append(self.flush())
        return out

    def poll(self, now=None):
        now = self.clock() if now is None else now
        if self.current and now - self.last_flush >= self.batch_timeout_s:      # ticker :109-118
            return [self.flush(now)]
        return []

    def flush(self, now=None):
        batch, self.current = self.current, []
        self.last_flush = self.clock() if now is None else now
        return batch


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
