// Go benchmark of the reference's own Engine.Step, to be run wherever a Go toolchain and a checkout
// of mitchelldurbincs/GeneralsReinforcementLearning exist (bench.py copies this file into the
// reference's internal/game/ as a _test.go file when GRL_REFERENCE_DIR is set and `go` is on PATH;
// it never runs in this repository's build image, which has no Go toolchain).
//
// The reference publishes no Step benchmark (internal/game/engine_bench_test.go covers stats and
// fog only); this one follows that file's conventions: package game, zerolog.Nop(), seeded
// math/rand, b.ResetTimer() after setup.  Workload = BASELINE.json's metric config: 20x20 boards,
// 4 players, fog of war on, random agents.  The action mix is the one bench.py's device agent
// uses (a player sits out with p = 0.1, moves half with p = 0.3, else uniform over the legal
// moves of Engine.GetLegalActionMask), so one op = one env-step of the same kind.
package game

import (
	"context"
	"math/rand"
	"testing"

	"github.com/mitchelldurbincs/GeneralsReinforcementLearning/internal/game/core"
	"github.com/rs/zerolog"
)

func gvecRandomActions(e *Engine, rng *rand.Rand, w int) []core.Action {
	var actions []core.Action
	st := e.GameState()
	for _, p := range st.Players {
		if !p.Alive || rng.Float32() < 0.1 {
			continue
		}
		mask := e.GetLegalActionMask(p.ID)
		n := 0
		for _, ok := range mask {
			if ok {
				n++
			}
		}
		if n == 0 {
			continue
		}
		k := rng.Intn(n)
		for idx, ok := range mask {
			if !ok {
				continue
			}
			if k == 0 {
				t, d := idx/4, idx%4 // rules/legal_moves.go:13-18: 0 up, 1 right, 2 down, 3 left
				x, y := t%w, t/w
				dx := []int{0, 1, 0, -1}[d]
				dy := []int{-1, 0, 1, 0}[d]
				actions = append(actions, &core.MoveAction{PlayerID: p.ID, FromX: x, FromY: y, ToX: x + dx, ToY: y + dy,
					MoveAll: rng.Float32() >= 0.3})
				break
			}
			k--
		}
	}
	return actions
}

func benchmarkEngineStep(b *testing.B, w, h, players int) {
	ctx := context.Background()
	rng := rand.New(rand.NewSource(1))
	newEngine := func(seed int64) *Engine {
		return NewEngine(ctx, GameConfig{Width: w, Height: h, Players: players, Rng: rand.New(rand.NewSource(seed)), Logger: zerolog.Nop()})
	}
	seed := int64(1)
	e := newEngine(seed)
	if e == nil {
		b.Fatal("NewEngine returned nil")
	}
	b.ResetTimer()
	for i := 0; i < b.N; i++ {
		if e.IsGameOver() { // auto-reset, like the vector env: the re-deal is outside the timed region
			b.StopTimer()
			seed++
			e = newEngine(seed)
			b.StartTimer()
		}
		acts := gvecRandomActions(e, rng, w)
		_ = e.Step(ctx, acts) // a move invalidated by a lower player id aborts the turn (engine.go:111-113): still one step
	}
}

func BenchmarkEngineStep20x20P4(b *testing.B) { benchmarkEngineStep(b, 20, 20, 4) }
func BenchmarkEngineStep15x15P2(b *testing.B) { benchmarkEngineStep(b, 15, 15, 2) }
func BenchmarkEngineStep10x10P2(b *testing.B) { benchmarkEngineStep(b, 10, 10, 2) }
