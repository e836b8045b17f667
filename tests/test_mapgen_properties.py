"""The map generator restated in oracle/ (ora_mapgen) against the RNG-independent assertions of the reference's
own generator tests (internal/game/mapgen/generator_test.go).  The reference's seed-specific counts (":82 22
mountains for seed 12345") depend on Go's math/rand stream and cannot be reproduced without the Go runtime:
distribution-level parity only.  The device generator is compared bit-for-bit with this
one in test_hip_parity.py::test_device_mapgen_matches_oracle_mapgen."""
import numpy as np
import pytest

import _oracle as O

NORMAL, GENERAL, CITY, MOUNTAIN = 0, 1, 2, 3


def gen(seed, env, w, h, p):
    rc, army, owner, typ = O.mapgen(seed, env, w, h, p)
    return rc, army.reshape(h, w), owner.reshape(h, w), typ.reshape(h, w)


@pytest.mark.parametrize("w,h,p", [(20, 15, 2), (20, 20, 4), (25, 25, 4), (10, 10, 1), (32, 32, 8)])
def test_default_config_properties(w, h, p):
    n = w * h
    for env in range(40):
        rc, army, owner, typ = gen(12345, env, w, h, p)
        assert rc == 0
        # TestPlaceMountains/BasicMountainPlacement :73-79: mountains are neutral with army 0;
        # DefaultMapConfig :25-33: N/50 veins of length 3..W/4
        m = typ == MOUNTAIN
        assert (army[m] == 0).all() and (owner[m] == -1).all()
        assert m.sum() <= (n // 50) * max(3, w // 4)
        # TestPlaceCities/BasicCityPlacement :166-174: N/CityRatio(20) cities, neutral, CityStartArmy (40)
        c = typ == CITY
        assert c.sum() == n // 20
        assert (army[c] == 40).all() and (owner[c] == -1).all()
        # TestPlaceGenerals/BasicGeneralPlacementAndSpacing :258-283 and FullIntegration :419-460:
        # one general per player, army 2, owned by that player, pairwise Manhattan distance >= spacing
        g = np.argwhere(typ == GENERAL)
        assert len(g) == p
        assert sorted(owner[typ == GENERAL].tolist()) == list(range(p))
        assert (army[typ == GENERAL] == 2).all()
        spacing = min(5, w // 2 + h // 2)  # DefaultMapConfig + clamp (:35-40)
        for i in range(len(g)):
            for j in range(i + 1, len(g)):
                assert abs(g[i] - g[j]).sum() >= spacing
        # every other tile is an empty neutral normal tile
        rest = typ == NORMAL
        assert (army[rest] == 0).all() and (owner[rest] == -1).all()


def test_small_board_spacing_is_clamped():
    # TestDefaultMapConfigClampsSpacingOnSmallBoards :35-48: 5x5, 2 players never fails
    for seed in range(50):
        rc, army, owner, typ = gen(seed, 0, 5, 5, 2)
        assert rc == 0
        g = np.argwhere(typ == GENERAL)
        assert len(g) == 2 and abs(g[0] - g[1]).sum() >= 4


def test_no_room_for_cities_or_veins_on_tiny_boards():
    # NoCitiesIfRatioIsTooHigh :177-192 / NoMountainVeins :85-100: N/20 == 0 cities, N/50 == 0 veins
    rc, army, owner, typ = gen(1, 0, 4, 4, 2)
    assert rc == 0
    assert (typ == CITY).sum() == 0 and (typ == MOUNTAIN).sum() == 0


def test_impossible_placement_reports_an_error():
    # PanicOnImpossibleSpacing :347-357: more generals than free tiles allow -> "unable to place general"
    rc, *_ = gen(1, 0, 1, 2, 3)
    assert rc != 0


def test_streams_are_independent_and_reproducible():
    a = gen(7, 3, 20, 20, 4)
    b = gen(7, 3, 20, 20, 4)
    c = gen(7, 4, 20, 20, 4)
    assert all(np.array_equal(x, y) for x, y in zip(a[1:], b[1:]))
    assert not np.array_equal(a[3], c[3])
