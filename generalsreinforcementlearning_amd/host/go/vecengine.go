// Package vecengine is the cgo shim a maintainer of
// mitchelldurbincs/GeneralsReinforcementLearning adds to call the MI355X batched
// turn engine (libgvec_hip.so, include/generals_vec.h) from Go.
//
// It keeps the reference's types at the boundary: actions are []core.Action
// (core.MoveAction), errors are the core sentinels, masks are []bool in the
// Engine.GetLegalActionMask order.  One VecEngine replaces B *game.Engine values
// (internal/game/engine.go:17-43); like Engine it is not goroutine-safe and is
// meant to be driven under the caller's lock (game_manager.go:576-602).
//
// WHERE THIS FILE GOES.  It imports .../internal/game/core (and vecengine_diff_test.go imports .../internal/game): Go
// only lets packages INSIDE the reference module import its internal/ tree, so these files cannot be built from this
// repository's directory.  Copy vecengine.go and vecengine_diff_test.go into the reference checkout as
//
//     <reference>/internal/game/vecengine/vecengine.go
//     <reference>/internal/game/vecengine/vecengine_diff_test.go
//
// and point cgo at this repository (GVEC = its checkout; the .so must have been built: csrc/build.py):
//
//     export CGO_CFLAGS="-I$GVEC/include"
//     export CGO_LDFLAGS="-L$GVEC/generalsreinforcementlearning_amd -lgvec_hip -Wl,-rpath,$GVEC/generalsreinforcementlearning_amd"
//     cd <reference> && go test ./internal/game/vecengine/        # needs an MI355X: there is no CPU fallback
//
// There are deliberately no `#cgo CFLAGS/LDFLAGS` directives below: relative ${SRCDIR} paths would only be right for one
// particular layout of the two checkouts.  NOT COMPILED IN THIS REPOSITORY'S CI: the build image has no Go toolchain.
package vecengine

/*
#include <stdlib.h>
#include "generals_vec.h"
*/
import "C"

import (
	"fmt"
	"unsafe"

	"github.com/mitchelldurbincs/GeneralsReinforcementLearning/internal/game/core"
)

// sentinel for each per-env code (core/errors.go:8-17; common.proto:39-48)
var sentinels = map[int32]error{
	1: core.ErrInvalidCoordinates, 2: core.ErrNotAdjacent, 3: core.ErrNotOwned, 4: core.ErrInsufficientArmy,
	5: core.ErrGameOver, 6: core.ErrInvalidPlayer, 7: core.ErrMoveToSelf, 8: core.ErrTargetIsMountain,
}

// Config mirrors game.GameConfig (engine.go:45-58) for a batch.
type Config struct {
	NumEnvs, Width, Height, Players int
	Device                          int
	Devices                         []int // non-empty: ONE engine over several GPUs (gvec_create_sharded), Device ignored
	FogOfWar                        bool // GameState.FogOfWarEnabled (engine_initializer.go:118: true)
	AutoReset                       bool
}

type VecEngine struct {
	h         *C.gvec_handle
	cfg       Config
	stride    int
	maskBytes int
	actions   []C.gvec_action // [B][Players]
	errs      []C.int32_t     // [B]
	maskBits  []C.uint8_t     // [B][Players][maskBytes]
	pinned    [3]unsafe.Pointer // the three slices above, page-locked (gvec_host_alloc)
}

func apiErr(rc C.int32_t, what string) error {
	if rc >= 0 {
		return nil
	}
	return fmt.Errorf("%s: gvec status %d: %s", what, int(rc), C.GoString(C.gvec_last_error()))
}

// NewVecEngine replaces game.NewEngine (engine.go:62-71) for B engines.
func NewVecEngine(cfg Config) (*VecEngine, error) {
	var c C.gvec_config
	C.gvec_config_default(&c)
	c.num_envs, c.max_width, c.max_height, c.max_players = C.int32_t(cfg.NumEnvs), C.int32_t(cfg.Width), C.int32_t(cfg.Height), C.int32_t(cfg.Players)
	c.device = C.int32_t(cfg.Device)
	if !cfg.FogOfWar {
		c.fog_of_war = 0
	}
	if cfg.AutoReset {
		c.auto_reset = 1
	}
	e := &VecEngine{cfg: cfg}
	if len(cfg.Devices) > 0 {
		// boards split into contiguous shards, one per listed device; every call below works unchanged on all B boards
		devs := make([]C.int32_t, len(cfg.Devices))
		for i, d := range cfg.Devices {
			devs[i] = C.int32_t(d)
		}
		if err := apiErr(C.gvec_create_sharded(&c, &devs[0], C.int32_t(len(devs)), &e.h), "gvec_create_sharded"); err != nil {
			return nil, err
		}
	} else if err := apiErr(C.gvec_create(&c, &e.h), "gvec_create"); err != nil {
		return nil, err
	}
	e.stride = int(C.gvec_tile_stride(e.h))
	e.maskBytes = int(C.gvec_mask_bytes(e.h))
	// The per-step buffers live in page-locked C memory (gvec_host_alloc): the 832-byte-per-board masks alone are 218 MB
	// per step at 262,144 boards - 4 ms over PCIe from pinned memory, 18 ms from Go-heap (pageable) slices.
	na, ne, nm := cfg.NumEnvs*cfg.Players, cfg.NumEnvs, cfg.NumEnvs*cfg.Players*e.maskBytes
	for i, bytes := range []int{na * int(unsafe.Sizeof(C.gvec_action{})), ne * 4, nm} {
		if err := apiErr(C.gvec_host_alloc(C.uint64_t(bytes), &e.pinned[i]), "gvec_host_alloc"); err != nil {
			e.Close()
			return nil, err
		}
	}
	e.actions = unsafe.Slice((*C.gvec_action)(e.pinned[0]), na)
	e.errs = unsafe.Slice((*C.int32_t)(e.pinned[1]), ne)
	e.maskBits = unsafe.Slice((*C.uint8_t)(e.pinned[2]), nm)
	return e, nil
}

func (e *VecEngine) Close() {
	if e.h != nil {
		C.gvec_destroy(e.h)
		e.h = nil
	}
	for i := range e.pinned {
		if e.pinned[i] != nil {
			C.gvec_host_free(e.pinned[i])
			e.pinned[i] = nil
		}
	}
}

// ResetBoards uploads one *core.Board per env (e.g. from mapgen.NewGenerator(...).GenerateMap(),
// engine_initializer.go:106-110) and runs performInitialSetup (engine_initializer.go:218-225).
func (e *VecEngine) ResetBoards(boards []*core.Board, players []int) error {
	n := len(boards)
	army := make([]C.int32_t, n*e.stride)
	owner := make([]C.int8_t, n*e.stride)
	typ := make([]C.uint8_t, n*e.stride)
	w, h, p := make([]C.int32_t, n), make([]C.int32_t, n), make([]C.int32_t, n)
	for i, b := range boards {
		w[i], h[i], p[i] = C.int32_t(b.W), C.int32_t(b.H), C.int32_t(players[i])
		for t, tile := range b.T { // row-major y*W+x, core/board.go:108
			army[i*e.stride+t] = C.int32_t(tile.Army)
			owner[i*e.stride+t] = C.int8_t(tile.Owner)
			typ[i*e.stride+t] = C.uint8_t(tile.Type)
		}
	}
	return apiErr(C.gvec_reset(e.h, nil, C.int32_t(n), &army[0], &owner[0], &typ[0], &w[0], &h[0], &p[0], C.GVEC_MEM_HOST), "gvec_reset")
}

// ResetGoSeeded builds env i's board on the GPU from Go's own math/rand stream: exactly the board
// game.NewEngine(ctx, game.GameConfig{Width, Height, Players, Rng: rand.New(rand.NewSource(seeds[i]))}) starts from
// (mapgen.DefaultMapConfig + GenerateMap, engine_initializer.go:106-110).  No board crosses PCIe.
func (e *VecEngine) ResetGoSeeded(seeds []int64) error {
	if len(seeds) != e.cfg.NumEnvs {
		return fmt.Errorf("ResetGoSeeded: %d seeds for %d envs", len(seeds), e.cfg.NumEnvs)
	}
	return apiErr(C.gvec_reset_go_seeded(e.h, (*C.int64_t)(unsafe.Pointer(&seeds[0])), nil, nil, nil), "gvec_reset_go_seeded")
}

func clamp8(v int) C.int8_t {
	if v < -128 {
		v = -128
	}
	if v > 127 {
		v = 127
	}
	return C.int8_t(v)
}

// Step is Engine.Step (engine.go:75) for every env: actions[env] holds that env's []core.Action
// for the turn (nil / missing player = no-op, converters.go:106-108).  The returned slice holds
// nil or the wrapped sentinel per env, in the form Engine.Step returns it (core/errors.go:35-40).
func (e *VecEngine) Step(actions [][]core.Action) ([]error, error) {
	for i := range e.actions {
		e.actions[i] = C.gvec_action{}
	}
	for env, list := range actions {
		for _, a := range list {
			m, ok := a.(*core.MoveAction)
			if !ok || m == nil {
				continue
			}
			if m.PlayerID < 0 || m.PlayerID >= e.cfg.Players {
				continue // action_processor.go:56-60: ignored
			}
			ga := &e.actions[env*e.cfg.Players+m.PlayerID]
			ga.from_x, ga.from_y, ga.to_x, ga.to_y = clamp8(m.FromX), clamp8(m.FromY), clamp8(m.ToX), clamp8(m.ToY)
			ga.flags = C.GVEC_ACT_VALID
			if !m.MoveAll {
				ga.flags |= C.GVEC_ACT_HALF
			}
		}
	}
	rc := C.gvec_step(e.h, &e.actions[0], &e.errs[0], &e.maskBits[0], C.GVEC_MEM_HOST)
	if err := apiErr(rc, "gvec_step"); err != nil {
		return nil, err
	}
	out := make([]error, len(e.errs))
	var turns []int32 // read back only when some env failed: the reference's message carries gs.Turn
	for i, code := range e.errs {
		if code == 0 {
			continue
		}
		if turns == nil {
			turns = make([]int32, len(e.errs))
			var v C.gvec_state_view
			v.turn = (*C.int32_t)(unsafe.Pointer(&turns[0]))
			if err := apiErr(C.gvec_read_state(e.h, 0, C.int32_t(len(e.errs)), &v, C.GVEC_MEM_HOST), "gvec_read_state"); err != nil {
				return nil, err
			}
		}
		s := sentinels[int32(code)]
		if code == 5 {
			// TurnProcessor.validateGameState (turn_processor.go:95-113): a finished engine refuses the turn.
			// (A naturally finished game trips the phase check first in Go and returns an unwrapped
			// "game is in ... phase" error; both are reported as ErrGameOver here.)
			out[i] = core.WrapGameStateError(int(turns[i]), "step", s)
			continue
		}
		// Engine.processActions wraps the first move error with the (already incremented) turn (engine.go:111-113),
		// TurnProcessor.processActionsPhase wraps that again (turn_processor.go:143-156).  The innermost
		// core.WrapActionError text (which move failed) is not reproduced: errors.Is sees the same sentinel.
		out[i] = core.WrapGameStateError(int(turns[i]), "action processing",
			core.WrapGameStateError(int(turns[i]), "processing actions", s))
	}
	return out, nil
}

// GetLegalActionMask is Engine.GetLegalActionMask(playerID) (engine.go:271-280) of one env,
// unpacked from the bits the last Step returned.
func (e *VecEngine) GetLegalActionMask(env, playerID, w, h int) []bool {
	mask := make([]bool, w*h*4)
	if playerID < 0 || playerID >= e.cfg.Players {
		return mask // engine.go:273-276
	}
	bits := e.maskBits[(env*e.cfg.Players+playerID)*e.maskBytes:]
	plane := e.maskBytes / 4 // four direction bit-planes (generals_vec.h, gvec_step)
	for t := 0; t < w*h; t++ {
		for d := 0; d < 4; d++ {
			mask[t*4+d] = bits[d*plane+(t>>3)]&(1<<uint(t&7)) != 0
		}
	}
	return mask
}

// State is the part of game.GameState (state.go:25-34) + Engine flags callers read after a Step.
type State struct {
	Army                       []int32
	Owner                      []int8
	Type, Visible              []uint8
	Changed, VisibilityChanged []uint8
	Turn                       []int32
	GameOver                   []uint8
	Winner                     []int8
	Alive                      []uint8
	ArmyCount, GeneralIdx      []int32
}

// GameState reads envs [begin, begin+n): GameState() / IsGameOver() / GetWinner() /
// GetChangedTiles() / GetVisibilityChangedTiles() (engine.go:197-298).
func (e *VecEngine) GameState(begin, n int) (*State, error) {
	s := &State{
		Army: make([]int32, n*e.stride), Owner: make([]int8, n*e.stride), Type: make([]uint8, n*e.stride),
		Visible: make([]uint8, n*e.stride), Changed: make([]uint8, n*e.stride), VisibilityChanged: make([]uint8, n*e.stride),
		Turn: make([]int32, n), GameOver: make([]uint8, n), Winner: make([]int8, n),
		Alive: make([]uint8, n*e.cfg.Players), ArmyCount: make([]int32, n*e.cfg.Players), GeneralIdx: make([]int32, n*e.cfg.Players),
	}
	var v C.gvec_state_view
	v.army = (*C.int32_t)(unsafe.Pointer(&s.Army[0]))
	v.owner = (*C.int8_t)(unsafe.Pointer(&s.Owner[0]))
	v._type = (*C.uint8_t)(unsafe.Pointer(&s.Type[0]))
	v.visible = (*C.uint8_t)(unsafe.Pointer(&s.Visible[0]))
	v.changed = (*C.uint8_t)(unsafe.Pointer(&s.Changed[0]))
	v.vis_changed = (*C.uint8_t)(unsafe.Pointer(&s.VisibilityChanged[0]))
	v.turn = (*C.int32_t)(unsafe.Pointer(&s.Turn[0]))
	v.done = (*C.uint8_t)(unsafe.Pointer(&s.GameOver[0]))
	v.winner = (*C.int8_t)(unsafe.Pointer(&s.Winner[0]))
	v.alive = (*C.uint8_t)(unsafe.Pointer(&s.Alive[0]))
	v.army_count = (*C.int32_t)(unsafe.Pointer(&s.ArmyCount[0]))
	v.general_idx = (*C.int32_t)(unsafe.Pointer(&s.GeneralIdx[0]))
	return s, apiErr(C.gvec_read_state(e.h, C.int32_t(begin), C.int32_t(n), &v, C.GVEC_MEM_HOST), "gvec_read_state")
}

// ComputePlayerVisibility is Engine.ComputePlayerVisibility(playerID) (visibility.go:153) for all envs.
func (e *VecEngine) ComputePlayerVisibility(playerID int) (visible, fog []uint8, err error) {
	visible = make([]uint8, e.cfg.NumEnvs*e.stride)
	fog = make([]uint8, e.cfg.NumEnvs*e.stride)
	rc := C.gvec_player_visibility(e.h, C.int32_t(playerID), (*C.uint8_t)(unsafe.Pointer(&visible[0])), (*C.uint8_t)(unsafe.Pointer(&fog[0])), C.GVEC_MEM_HOST)
	return visible, fog, apiErr(rc, "gvec_player_visibility")
}

// GatherExperienceRecords is the sharded engine's hand-off to the process that feeds StreamAggregator
// (internal/grpc/gameserver/stream_aggregator.go:75-155): after ExperienceBegin() + Step(), every GPU writes the compact
// experience records of its envs [shardEnvBegin, shardEnvBegin+n) and copies them into one host slab over its own PCIe
// link.  Record layout: include/generals_vec.h "experience records"; expand with the rules of experience.py decode_records.
func (e *VecEngine) ExperienceBegin() error { return apiErr(C.gvec_experience_begin(e.h), "gvec_experience_begin") }

func (e *VecEngine) GatherExperienceRecords(shardEnvBegin, n, envIDBase int) ([]byte, error) {
	shards := int(C.gvec_num_shards(e.h))
	rec := int(C.gvec_experience_record_bytes(e.h))
	out := make([]byte, shards*n*rec)
	rc := C.gvec_gather_experience_records(e.h, C.int32_t(shardEnvBegin), C.int32_t(n), C.int32_t(envIDBase), C.GVEC_MEM_HOST, 0, unsafe.Pointer(&out[0]))
	return out, apiErr(rc, "gvec_gather_experience_records")
}

// StreamDeltas is what gameInstance.broadcastUpdates needs of the boards after a turn (server.go:612-777), for the stream
// of `playerID`, for every game at once: kind[env] = 1 -> send a GameStateDelta whose tile updates are
// updates[offset[env]:offset[env+1]] (bits 0-15 tile index, 16-17 core tile type after the fog rules, 18 visible,
// 19 fog_of_war, 20-23 owner+1, 32-63 army); kind[env] = 2 -> createStreamUpdate falls back to the full state
// (convertGameStateToProto over GameState(env, 1)).  PlayerUpdates come from GameState's per-player fields.
func (e *VecEngine) StreamDeltas(playerID int) (kind []uint8, offset []int64, updates []uint64, err error) {
	B := e.cfg.NumEnvs
	capacity := B * int(C.gvec_stream_delta_cap(e.h))
	kind, offset, updates = make([]uint8, B), make([]int64, B+1), make([]uint64, capacity)
	var total C.int64_t
	rc := C.gvec_stream_deltas_packed(e.h, C.int32_t(playerID), 0, (*C.uint8_t)(unsafe.Pointer(&kind[0])), (*C.int64_t)(unsafe.Pointer(&offset[0])),
		(*C.uint64_t)(unsafe.Pointer(&updates[0])), C.int64_t(capacity), &total)
	return kind, offset, updates[:int(total)], apiErr(rc, "gvec_stream_deltas_packed")
}

