"""Proto adapters: the reference's gRPC surface (proto/game/v1, proto/experience/v1) served from the batched engine.

The reference keeps its host side in Go; what a Go maintainer adds there is host/go/vecengine.go.  This module is
the same glue in Python (SURVEY 8f n3 - the Go toolchain is absent from the build image, the wire format is not):

  * game_state(...)      = Server.convertGameStateToProto (internal/grpc/gameserver/server.go:526-610): fog rules per
                           viewing player (:556-582), PlayerState fields (:529-553), winner, action mask, phase.
  * stream_update(...)   = gameInstance.createStreamUpdate (server.go:632-777): a GameStateDelta iff
                           0 < |ChangedTiles| + |VisibilityChangedTiles| < boardSize / 5 (:640-644), else the full state.
  * experience(...)      = the experiencepb.Experience SimpleCollector.OnStateTransition builds (collector.go:58-83).
  * experience_batch(..) = ExperienceService.StreamExperienceBatches' framing (experience_service.go:340-349).

Message classes: the wire schema (package, message, field names / numbers / types of the three .proto files) is
declared below and turned into classes in a PRIVATE descriptor pool, so this module neither imports nor collides with
the reference's generated stubs (python/generals_pb); tests/test_wire.py checks in the build container that both
describe the same wire format (field tables and cross-parsing) - nothing of the reference travels to a GPU box.
"""
import numpy as np
from google.protobuf import descriptor_pb2, descriptor_pool, message_factory, timestamp_pb2

_F = descriptor_pb2.FieldDescriptorProto
_T = {"int32": _F.TYPE_INT32, "int64": _F.TYPE_INT64, "bool": _F.TYPE_BOOL, "string": _F.TYPE_STRING, "float": _F.TYPE_FLOAT}

# (field name, number, type, repeated) - type: scalar name, ".pkg.Message", "enum:.pkg.Enum", or ("map", key, value)
_SCHEMA = {
    "generals/common/v1/common.proto": {
        "package": "generals.common.v1", "deps": [],
        "enums": {
            "TileType": ["TILE_TYPE_UNSPECIFIED", "TILE_TYPE_NORMAL", "TILE_TYPE_GENERAL", "TILE_TYPE_CITY", "TILE_TYPE_MOUNTAIN"],
            "PlayerStatus": ["PLAYER_STATUS_UNSPECIFIED", "PLAYER_STATUS_ACTIVE", "PLAYER_STATUS_ELIMINATED", "PLAYER_STATUS_DISCONNECTED"],
            "GameStatus": ["GAME_STATUS_UNSPECIFIED", "GAME_STATUS_WAITING", "GAME_STATUS_IN_PROGRESS", "GAME_STATUS_FINISHED",
                           "GAME_STATUS_CANCELLED"],
            "GamePhase": ["GAME_PHASE_UNSPECIFIED", "GAME_PHASE_INITIALIZING", "GAME_PHASE_LOBBY", "GAME_PHASE_STARTING", "GAME_PHASE_RUNNING",
                          "GAME_PHASE_PAUSED", "GAME_PHASE_ENDING", "GAME_PHASE_ENDED", "GAME_PHASE_ERROR", "GAME_PHASE_RESET"],
        },
        "messages": {"Coordinate": [("x", 1, "int32", False), ("y", 2, "int32", False)]},
    },
    "generals/game/v1/game.proto": {
        "package": "generals.game.v1", "deps": ["generals/common/v1/common.proto", "google/protobuf/timestamp.proto"],
        "enums": {},
        "messages": {
            "Tile": [("type", 1, "enum:.generals.common.v1.TileType", False), ("owner_id", 2, "int32", False), ("army_count", 3, "int32", False),
                     ("visible", 4, "bool", False), ("fog_of_war", 5, "bool", False)],
            "Board": [("width", 1, "int32", False), ("height", 2, "int32", False), ("tiles", 3, ".generals.game.v1.Tile", True)],
            "PlayerState": [("id", 1, "int32", False), ("name", 2, "string", False), ("status", 3, "enum:.generals.common.v1.PlayerStatus", False),
                            ("army_count", 4, "int32", False), ("tile_count", 5, "int32", False),
                            ("general_position", 6, ".generals.common.v1.Coordinate", False), ("color", 7, "string", False)],
            "GameState": [("game_id", 1, "string", False), ("status", 2, "enum:.generals.common.v1.GameStatus", False), ("turn", 3, "int32", False),
                          ("board", 4, ".generals.game.v1.Board", False), ("players", 5, ".generals.game.v1.PlayerState", True),
                          ("winner_id", 6, "int32", False), ("started_at", 7, ".google.protobuf.Timestamp", False),
                          ("updated_at", 8, ".google.protobuf.Timestamp", False), ("action_mask", 9, "bool", True),
                          ("current_phase", 10, "enum:.generals.common.v1.GamePhase", False)],
            "TileUpdate": [("position", 1, ".generals.common.v1.Coordinate", False), ("tile", 2, ".generals.game.v1.Tile", False)],
            "PlayerUpdate": [("player_id", 1, "int32", False), ("state", 2, ".generals.game.v1.PlayerState", False)],
            "GameStateDelta": [("turn", 1, "int32", False), ("tile_updates", 2, ".generals.game.v1.TileUpdate", True),
                               ("player_updates", 3, ".generals.game.v1.PlayerUpdate", True)],
            # GameUpdate.update is a oneof {full_state = 1, delta = 2, event = 3}; events are not produced here
            "GameUpdate": [("full_state", 1, ".generals.game.v1.GameState", False, "update"), ("delta", 2, ".generals.game.v1.GameStateDelta", False, "update"),
                           ("timestamp", 4, ".google.protobuf.Timestamp", False)],
        },
    },
    "generals/experience/v1/experience.proto": {
        "package": "generals.experience.v1", "deps": ["google/protobuf/timestamp.proto"],
        "enums": {},
        "messages": {
            "TensorState": [("shape", 1, "int32", True), ("data", 2, "float", True)],
            "Experience": [("experience_id", 1, "string", False), ("game_id", 2, "string", False), ("player_id", 3, "int32", False),
                           ("turn", 4, "int32", False), ("state", 5, ".generals.experience.v1.TensorState", False), ("action", 6, "int32", False),
                           ("reward", 7, "float", False), ("next_state", 8, ".generals.experience.v1.TensorState", False), ("done", 9, "bool", False),
                           ("action_mask", 10, "bool", True), ("collected_at", 11, ".google.protobuf.Timestamp", False),
                           ("metadata", 12, ("map", "string", "string"), True)],
            "ExperienceBatch": [("experiences", 1, ".generals.experience.v1.Experience", True), ("batch_id", 2, "int32", False),
                                ("stream_id", 3, "string", False), ("created_at", 4, ".google.protobuf.Timestamp", False),
                                ("metadata", 5, ("map", "string", "string"), True)],
        },
    },
}


def _build_pool():
    pool = descriptor_pool.DescriptorPool()
    pool.AddSerializedFile(timestamp_pb2.DESCRIPTOR.serialized_pb)
    for fname, spec in _SCHEMA.items():
        fd = descriptor_pb2.FileDescriptorProto(name=fname, package=spec["package"], syntax="proto3", dependency=spec["deps"])
        for ename, values in spec["enums"].items():
            e = fd.enum_type.add(name=ename)
            for i, v in enumerate(values):
                e.value.add(name=v, number=i)
        for mname, fields in spec["messages"].items():
            m = fd.message_type.add(name=mname)
            oneofs = {}
            for f in fields:
                name, number, typ, repeated = f[:4]
                fp = m.field.add(name=name, number=number, label=_F.LABEL_REPEATED if repeated else _F.LABEL_OPTIONAL)
                if isinstance(typ, tuple):      # map<k, v>: a nested <Name>Entry message with map_entry set
                    entry = "".join(p.capitalize() for p in name.split("_")) + "Entry"
                    nm = m.nested_type.add(name=entry)
                    nm.options.map_entry = True
                    nm.field.add(name="key", number=1, label=_F.LABEL_OPTIONAL, type=_T[typ[1]])
                    nm.field.add(name="value", number=2, label=_F.LABEL_OPTIONAL, type=_T[typ[2]])
                    fp.type, fp.type_name = _F.TYPE_MESSAGE, f".{spec['package']}.{mname}.{entry}"
                elif typ.startswith("enum:"):
                    fp.type, fp.type_name = _F.TYPE_ENUM, typ[5:]
                elif typ.startswith("."):
                    fp.type, fp.type_name = _F.TYPE_MESSAGE, typ
                else:
                    fp.type = _T[typ]
                if len(f) > 4:
                    if f[4] not in oneofs:
                        oneofs[f[4]] = len(m.oneof_decl)
                        m.oneof_decl.add(name=f[4])
                    fp.oneof_index = oneofs[f[4]]
        pool.Add(fd)
    return pool


POOL = _build_pool()


def _cls(full_name):
    return message_factory.GetMessageClass(POOL.FindMessageTypeByName(full_name))


Coordinate = _cls("generals.common.v1.Coordinate")
Tile, Board, PlayerState, GameState = (_cls("generals.game.v1." + n) for n in ("Tile", "Board", "PlayerState", "GameState"))
TileUpdate, PlayerUpdate, GameStateDelta, GameUpdate = (_cls("generals.game.v1." + n) for n in ("TileUpdate", "PlayerUpdate", "GameStateDelta", "GameUpdate"))
TensorState, Experience, ExperienceBatch = (_cls("generals.experience.v1." + n) for n in ("TensorState", "Experience", "ExperienceBatch"))

TILE_TYPE = {0: 1, 1: 2, 2: 3, 3: 4}               # convertTileType (converters.go:15-28): core -> TILE_TYPE_*
PLAYER_ACTIVE, PLAYER_ELIMINATED = 1, 2
PHASE_RUNNING, PHASE_ENDED = 4, 7
STATUS_IN_PROGRESS, STATUS_FINISHED = 2, 3          # mapPhaseToStatus (converters.go:84-102)
NUM_CHANNELS = 9


def player_color(i):
    """generatePlayerColor (converters.go:135-137)."""
    return "#%06X" % (i * 0x333333)


def _tile(st, e, t, vis, fog):
    """One proto Tile with the fog rules of server.go:556-582 (== :664-689 for deltas)."""
    visible, fogged = bool(vis[e, t]), bool(fog[e, t])
    typ, owner, army = TILE_TYPE.get(int(st["type"][e, t]), 0), int(st["owner"][e, t]), int(st["army"][e, t])
    if not visible and not fogged:          # completely hidden
        typ, owner, army = 1, -1, 0
    elif fogged and not visible:            # in fog: the type shows, the current state does not
        owner, army = -1, 0
    return Tile(type=typ, owner_id=owner, army_count=army, visible=visible, fog_of_war=fogged)


def _player_state(st, e, p, name, viewer, delta):
    w = int(st["width"][e])
    alive = bool(st["alive"][e, p])
    ps = PlayerState(id=p, name=name, status=PLAYER_ACTIVE if alive else PLAYER_ELIMINATED, army_count=int(st["army_count"][e, p]),
                     tile_count=int(st["tile_count"][e, p]), color=player_color(p))
    gi = int(st["general_idx"][e, p])
    # full state (server.go:547-555): shown if eliminated or the general tile is owned by the viewer;
    # delta (:741-752): shown only when eliminated
    show = gi >= 0 and (not alive or (not delta and int(st["owner"][e, gi]) == viewer))
    if show:
        ps.general_position.x, ps.general_position.y = gi % w, gi // w
    return ps


STATE_FIELDS = ("army", "owner", "type", "changed", "vis_changed", "turn", "done", "winner", "width", "height", "players", "alive", "army_count",
                "tile_count", "general_idx")


def game_state(st, vis, fog, legal_mask, e, viewer, game_id="", names=None):
    """Server.convertGameStateToProto for env `e` as seen by player `viewer`.
    st: VecEngine.game_state(fields=STATE_FIELDS); vis / fog: VecEngine.compute_player_visibility(viewer);
    legal_mask: bool[W*H*4] = Engine.GetLegalActionMask(viewer) (VecEngine.get_legal_action_mask / unpack_legal_bits)."""
    w, h, P = int(st["width"][e]), int(st["height"][e]), int(st["players"][e])
    names = names or [f"player{p}" for p in range(P)]
    done = bool(st["done"][e])
    phase = PHASE_ENDED if done else PHASE_RUNNING      # the only phases a stepping engine is in (states/phases.go:69-71)
    gs = GameState(game_id=game_id, status=STATUS_FINISHED if done else STATUS_IN_PROGRESS, turn=int(st["turn"][e]),
                   winner_id=int(st["winner"][e]) if done else -1, current_phase=phase)
    gs.board.width, gs.board.height = w, h
    gs.board.tiles.extend(_tile(st, e, t, vis, fog) for t in range(w * h))
    gs.players.extend(_player_state(st, e, p, names[p], viewer, delta=False) for p in range(P))
    gs.action_mask.extend(bool(v) for v in legal_mask)
    return gs


def stream_update(st, vis, fog, legal_mask, e, viewer, game_id="", names=None):
    """gameInstance.createStreamUpdate (server.go:632-777) for env e / player viewer -> GameUpdate."""
    w, h, P = int(st["width"][e]), int(st["height"][e]), int(st["players"][e])
    n = w * h
    changed = np.flatnonzero(st["changed"][e, :n])
    vchanged = np.flatnonzero(st["vis_changed"][e, :n])
    total = len(changed) + len(vchanged)               # :636 - a tile in both sets counts twice
    up = GameUpdate()
    up.timestamp.GetCurrentTime()
    if 0 < total < n // 5:                              # :640-644 (integer division)
        names = names or [f"player{p}" for p in range(P)]
        d = up.delta
        d.turn = int(st["turn"][e])
        seen = set()
        for t in list(changed) + list(vchanged):        # changed tiles first, then the visibility-only ones (:650-722)
            t = int(t)
            if t in seen:
                continue
            seen.add(t)
            tu = d.tile_updates.add()
            tu.position.x, tu.position.y = t % w, t // w
            tu.tile.CopyFrom(_tile(st, e, t, vis, fog))
        for p in range(P):                              # :725-760
            pu = d.player_updates.add(player_id=p)
            pu.state.CopyFrom(_player_state(st, e, p, names[p], viewer, delta=True))
    else:
        up.full_state.CopyFrom(game_state(st, vis, fog, legal_mask, e, viewer, game_id, names))
    return up


def stream_update_from_delta(st, kind, count, updates, e, viewer, names=None):
    """The GameUpdate of env e from gvec_stream_deltas' outputs (VecEngine.stream_deltas(viewer)) and the per-player fields
    of VecEngine.game_state: the delta itself, or None when the server would send the full state (kind 2: build it with
    `game_state`).  Equal to `stream_update` message for message (tests/test_wire.py)."""
    if int(kind[e]) != 1:
        return None
    w, P = int(st["width"][e]), int(st["players"][e])
    names = names or [f"player{p}" for p in range(P)]
    up = GameUpdate()
    up.timestamp.GetCurrentTime()
    d = up.delta
    d.turn = int(st["turn"][e])
    for u in updates[e, : int(count[e])]:
        u = int(u)
        t, typ, owner = u & 0xFFFF, (u >> 16) & 3, ((u >> 20) & 0xF) - 1
        army = (u >> 32) & 0xFFFFFFFF
        army = army - (1 << 32) if army >= 1 << 31 else army
        tu = d.tile_updates.add()
        tu.position.x, tu.position.y = t % w, t // w
        tu.tile.CopyFrom(Tile(type=TILE_TYPE[typ], owner_id=owner, army_count=army, visible=bool((u >> 18) & 1), fog_of_war=bool((u >> 19) & 1)))
    for p in range(P):
        pu = d.player_updates.add(player_id=p)
        pu.state.CopyFrom(_player_state(st, e, p, names[p], viewer, delta=True))
    return up


def full_state_from_tiles(st, tiles, legal_mask, e, viewer, game_id="", names=None):
    """convertGameStateToProto for env e from the packed tiles gvec_stream_deltas_packed(full_tiles=1) delivers for an env
    of kind 2 (all W*H tiles, ascending, fog rules applied): `game_state` without a board read-back."""
    w, h, P = int(st["width"][e]), int(st["height"][e]), int(st["players"][e])
    names = names or [f"player{p}" for p in range(P)]
    done = bool(st["done"][e])
    gs = GameState(game_id=game_id, status=STATUS_FINISHED if done else STATUS_IN_PROGRESS, turn=int(st["turn"][e]),
                   winner_id=int(st["winner"][e]) if done else -1, current_phase=PHASE_ENDED if done else PHASE_RUNNING)
    gs.board.width, gs.board.height = w, h
    for u in tiles:
        u = int(u)
        army = (u >> 32) & 0xFFFFFFFF
        gs.board.tiles.add(type=TILE_TYPE[(u >> 16) & 3], owner_id=((u >> 20) & 0xF) - 1, army_count=army - (1 << 32) if army >= 1 << 31 else army,
                           visible=bool((u >> 18) & 1), fog_of_war=bool((u >> 19) & 1))
    gs.players.extend(_player_state(st, e, p, names[p], viewer, delta=False) for p in range(P))
    gs.action_mask.extend(bool(v) for v in legal_mask)
    return gs


def experience(d, collector_version="1.0.0"):
    """One experience dict (VecExperienceCollector.as_dicts / decode_records) -> experiencepb.Experience as
    SimpleCollector.OnStateTransition fills it (collector.go:58-83)."""
    state, nxt = np.asarray(d["state"], np.float32), np.asarray(d["next_state"], np.float32)
    x = Experience(experience_id=str(d["experience_id"]), game_id=str(d["game_id"]), player_id=int(d["player_id"]), turn=int(d["turn"]),
                   action=int(d["action"]), reward=float(d["reward"]), done=bool(d["done"]))
    x.state.shape.extend(int(v) for v in state.shape)
    x.state.data.extend(state.ravel().tolist())
    x.next_state.shape.extend(int(v) for v in nxt.shape)
    x.next_state.data.extend(nxt.ravel().tolist())
    x.action_mask.extend(bool(v) for v in np.asarray(d["action_mask"]).ravel())
    x.collected_at.GetCurrentTime()
    x.metadata["collector_version"] = collector_version
    return x


def experience_batch(experiences, batch_id, stream_id):
    """ExperienceService.StreamExperienceBatches' batch message (experience_service.go:340-349)."""
    b = ExperienceBatch(batch_id=int(batch_id), stream_id=str(stream_id))
    b.experiences.extend(experiences)
    b.created_at.GetCurrentTime()
    b.metadata["batch_size"] = str(len(experiences))
    return b
