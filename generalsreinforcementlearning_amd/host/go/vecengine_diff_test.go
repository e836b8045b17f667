// Differential test against the real Go engine, for hosts that have both a Go toolchain and an
// MI355X: steps B reference engines (internal/game) and one VecEngine with identical boards and
// actions and compares every tile after every turn.  Not run in this repository's CI (no Go here).
// Lives next to vecengine.go INSIDE the reference module (<reference>/internal/game/vecengine/): it imports the
// reference's internal packages, which Go forbids from any other module - see vecengine.go's header for the
// destination and the CGO_CFLAGS / CGO_LDFLAGS to export.
package vecengine

import (
	"context"
	"math/rand"
	"testing"

	"github.com/mitchelldurbincs/GeneralsReinforcementLearning/internal/game"
	"github.com/mitchelldurbincs/GeneralsReinforcementLearning/internal/game/core"
	"github.com/rs/zerolog"
)

func TestVecEngineMatchesGoEngine(t *testing.T) {
	const B, W, H, P, turns = 256, 20, 20, 4, 300
	ctx := context.Background()
	ref := make([]*game.Engine, B)
	boards := make([]*core.Board, B)
	players := make([]int, B)
	for i := range ref {
		ref[i] = game.NewEngine(ctx, game.GameConfig{Width: W, Height: H, Players: P, Rng: rand.New(rand.NewSource(int64(i + 1))), Logger: zerolog.Nop()})
		boards[i] = ref[i].GameState().Board.Clone()
		players[i] = P
	}
	vec, err := NewVecEngine(Config{NumEnvs: B, Width: W, Height: H, Players: P, FogOfWar: true})
	if err != nil {
		t.Fatal(err)
	}
	defer vec.Close()
	// The boards are NOT uploaded: the library regenerates them on the GPU from the same seeds with its restatement of
	// Go's math/rand (gvec_reset_go_seeded) - so turn 0 already compares the reference's mapgen with the device's.
	// (vec.ResetBoards(boards, players) is the upload path, for boards that come from anywhere else.)
	_ = boards
	_ = players
	seeds := make([]int64, B)
	for i := range seeds {
		seeds[i] = int64(i + 1)
	}
	if err := vec.ResetGoSeeded(seeds); err != nil {
		t.Fatal(err)
	}
	if st0, err := vec.GameState(0, B); err != nil {
		t.Fatal(err)
	} else {
		for i := range ref {
			for tIdx, tile := range ref[i].GameState().Board.T {
				k := i*W*H + tIdx
				if int(st0.Army[k]) != tile.Army || int(st0.Owner[k]) != tile.Owner || int(st0.Type[k]) != int(tile.Type) {
					t.Fatalf("turn 0 env %d (seed %d) tile %d: go=%+v vec army=%d owner=%d type=%d", i, seeds[i], tIdx, tile, st0.Army[k], st0.Owner[k], st0.Type[k])
				}
			}
		}
	}
	rng := rand.New(rand.NewSource(99))
	for turn := 0; turn < turns; turn++ {
		acts := make([][]core.Action, B)
		for i := range ref {
			if ref[i].IsGameOver() {
				continue
			}
			acts[i] = game.GenerateRandomActions(ref[i], rng) // demo_helpers.go:12-62
		}
		errs, err := vec.Step(acts)
		if err != nil {
			t.Fatal(err)
		}
		for i := range ref {
			refErr := ref[i].Step(ctx, acts[i])
			if (refErr == nil) != (errs[i] == nil) {
				t.Fatalf("turn %d env %d: error mismatch: go=%v vec=%v", turn, i, refErr, errs[i])
			}
		}
		st, err := vec.GameState(0, B)
		if err != nil {
			t.Fatal(err)
		}
		for i := range ref {
			gs := ref[i].GameState()
			for tIdx, tile := range gs.Board.T {
				k := i*W*H + tIdx
				if int(st.Army[k]) != tile.Army || int(st.Owner[k]) != tile.Owner || uint32(st.Visible[k]) != tile.VisibleBitfield {
					t.Fatalf("turn %d env %d tile %d: go=%+v vec army=%d owner=%d vis=%d", turn, i, tIdx, tile, st.Army[k], st.Owner[k], st.Visible[k])
				}
			}
			for p := 0; p < P; p++ {
				if (st.Alive[i*P+p] != 0) != gs.Players[p].Alive || int(st.ArmyCount[i*P+p]) != gs.Players[p].ArmyCount {
					t.Fatalf("turn %d env %d player %d stats mismatch", turn, i, p)
				}
			}
		}
	}
}
