"""TEST INFRASTRUCTURE: restatement-free properties of one engine turn, in plain numpy.

The reference holds no asserting test for aborted turns, list desync (H6), stale-list production (H7) or
the incremental stats rule; the oracle's C restatement is otherwise their only witness.  These checks re-derive what a
turn must have done from the state BEFORE it, the actions and the state AFTER it, using nothing but the Go rules as
written (file:line cited per check) - so they hold for the oracle AND the HIP engine independently of each other.
Only turns whose bookkeeping is unambiguous from outside are checked for a given property (e.g. no elimination)."""
import numpy as np

GROWTH_INTERVAL = 25          # internal/config/config.go:209


def _planes(st, e, n, P):
    own = np.stack([st["owner"][e, :n] == p for p in range(P)])
    lst = np.stack([st["listed"][e, :n] == p for p in range(P)])
    return own, lst


def check_turn(before, acts, err, after, e, counters):
    """Checks env e of a batch; `counters` is a dict the caller accumulates coverage in."""
    w, h, P = int(before["width"][e]), int(before["height"][e]), int(before["players"][e])
    n = w * h
    if before["done"][e]:
        # turn_processor.go:95-113: a finished engine refuses the turn and stays as it was
        assert err[e] == 5
        for f in ("army", "owner", "listed", "visible", "turn", "alive", "army_count"):
            assert np.array_equal(before[f][e], after[f][e]), f
        counters["frozen"] += 1
        return
    assert after["turn"][e] == before["turn"][e] + 1, "initializeTurn increments Turn even when the turn aborts (turn_processor.go:124-135)"
    own_b, lst_b = _planes(before, e, n, P)
    own_a, lst_a = _planes(after, e, n, P)
    army_b, army_a = before["army"][e, :n].astype(np.int64), after["army"][e, :n].astype(np.int64)
    typ = after["type"][e, :n]
    assert np.array_equal(before["type"][e, :n], typ), "a capture never changes Tile.Type (movement.go:38, H3)"
    alive_b, alive_a = before["alive"][e, :P].astype(bool), after["alive"][e, :P].astype(bool)
    # the moves that reach ApplyMoveAction: present, PlayerID valid, Alive as last written (action_processor.go:56-60)
    moves = [(p, int(acts[e, p]["from_y"]) * w + int(acts[e, p]["from_x"]), int(acts[e, p]["to_y"]) * w + int(acts[e, p]["to_x"]))
             for p in range(P) if (acts[e, p]["flags"] & 1) and alive_b[p]]
    gen_captured = any(typ[t] == 1 and before["owner"][e, t] not in (-1, p) and after["owner"][e, t] != before["owner"][e, t]
                       for p, f, t in moves if 0 <= t < n)
    eliminated = gen_captured or (alive_b & ~alive_a).any()
    C = after["changed"][e, :n].astype(bool)
    if err[e] != 0:
        counters["aborted"] += 1
        if not eliminated:
            # engine.go:111-113 -> turn_processor.go:55-57 (H5): no production, no end-of-turn stats, no game-over check
            assert np.array_equal(lst_a, lst_b), "an aborted turn leaves OwnedTiles as they were (H6)"
            assert np.array_equal(before["army_count"][e], after["army_count"][e]) and np.array_equal(alive_b, alive_a)
            touched = np.zeros(n, bool)
            for p, f, t in moves:
                for x in (f, t):
                    if 0 <= x < n:
                        touched[x] = True
            assert np.array_equal(army_a[~touched], army_b[~touched]), "no production on an aborted turn"
            assert not (C & ~touched).any()
            counters["aborted_checked"] += 1
        return
    # ---- a completed turn -------------------------------------------------------------------------------------
    for p in range(P if C.any() else 0):                   # stats.go:33-63 / 90-144: what every pass leaves behind (|C| = 0: skipped)
        assert after["army_count"][e, p] == army_a[lst_a[p]].sum(), "ArmyCount = sum of Tile.Army over OwnedTiles"
        assert after["tile_count"][e, p] == lst_a[p].sum()
        assert alive_a[p] == bool((lst_a[p] & (typ == 1)).any()), "Alive <=> a general is listed (stats.go:52-54,133-135)"
    assert (lst_a & ~own_a).sum() == 0, "a pass never lists a tile its player does not own (stats.go:97)"
    if P > 1:
        assert bool(after["done"][e]) == (alive_a.sum() <= 1), "rules/win_conditions.go:21-57"
    if eliminated:
        counters["eliminations"] += 1
        return                                             # two stats passes and a turnover: not reconstructible from outside
    assert all(0 <= f < n and 0 <= t < n for _, f, t in moves), "a completed turn applied every submitted move"
    move_tiles = np.zeros(n, bool)
    for _, f, t in moves:
        move_tiles[f] = move_tiles[t] = True
    # production (production_manager.go:26-101, H7): the lists of alive players AS OF THE LAST STATS PASS - here the
    # turn's start - produce, whoever owns the tile by now; general / city every turn, normal tiles every 25th
    listed_alive = np.zeros(n, bool)
    for p in range(P):
        if alive_b[p]:
            listed_alive |= lst_b[p]
    turn = int(after["turn"][e])
    prod = listed_alive & ((typ == 1) | (typ == 2) | ((typ == 0) & (turn % GROWTH_INTERVAL == 0)))
    quiet = ~move_tiles
    assert np.array_equal(army_a[quiet] - army_b[quiet], prod[quiet].astype(np.int64)), "production follows the stale list (H7)"
    assert np.array_equal(C, move_tiles | prod), "ChangedTiles = moved tiles + produced tiles (movement.go:57-60, production_manager.go:59-61)"
    tiles_used = [x for _, f, t in moves for x in (f, t)]
    if len(set(tiles_used)) == len(tiles_used):            # no tile shared by two moves: captures are visible from outside
        captured = np.zeros(n, bool)
        for p, f, t in moves:
            if after["owner"][e, t] == p and before["owner"][e, t] != p:
                captured[t] = True
        assert np.array_equal(after["vis_changed"][e, :n].astype(bool), captured), "VisibilityChangedTiles = captured tiles (action_processor.go:84-86)"
    # the list rule of the one stats pass this turn ran (stats.go:10-14,20-21,33-49,90-130; H6)
    nc = int(C.sum())
    full = nc > n // 5
    want = lst_b if nc == 0 else (own_a if full else own_a & (lst_b | C[None, :]))   # |C| = 0: the pass is skipped
    assert np.array_equal(lst_a, want), f"OwnedTiles after a {'full' if full else 'incremental'} pass"
    counters["full" if full else "incremental"] += 1
    if (lst_b != own_b).any():
        counters["desynced_start"] += 1
    # moves (movement.go:40-86) on tiles that exactly one move touched
    for p, f, t in moves:
        if sum((f in (a, b)) + (t in (a, b)) for _, a, b in moves) != 2:
            continue                                       # another move shares a tile: order-dependent, left to the lock-step tests
        half = bool(acts[e, p]["flags"] & 2)
        fa, ta = int(army_b[f]), int(army_b[t])
        k = max(fa // 2, 1) if half else fa - 1
        pf, pt = int(prod[f]), int(prod[t])
        assert army_a[f] == fa - k + pf
        if before["owner"][e, t] == p:
            assert army_a[t] == ta + k + pt and after["owner"][e, t] == p
        elif k > ta:
            assert army_a[t] == k - ta + pt and after["owner"][e, t] == p
        else:
            assert army_a[t] == ta - k + pt and after["owner"][e, t] == before["owner"][e, t], "ties favour the defender"
        counters["moves"] += 1


def new_counters():
    return {k: 0 for k in ("frozen", "aborted", "aborted_checked", "eliminations", "full", "incremental", "desynced_start", "moves")}
