// gvec_launch.hpp — host-visible launchers of the HIP kernels in gvec_kernels.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gvec_device.hpp"

namespace gvec {

// which compiled (MAXP, NSLOT) variant serves a handle
struct Variant {
  int maxp;   // 2, 4, 8
  int nslot;  // 1, 2, 4, 7, 10, 16  (64-tile slots: ceil(max_w*max_h / 64) rounded up to a built size)
};
bool pick_variant(int max_players, int tile_stride, Variant* out);

struct ImportArgs {
  uint32_t* hdr;
  uint32_t* rows;
  uint32_t* army16;  // narrow / wide army storage of the destination (gvec_device.hpp "army storage")
  int32_t* army32;
  const int32_t* env_ids;  // [n] or null (env = dst_begin + i)
  int32_t dst_begin, n;
  // source planes, [n][stride] / [n][max_p] / [n]; null = keep current value
  const int32_t* s_army;
  const int8_t* s_owner;
  const uint8_t* s_type;
  const uint8_t* s_visible;
  const int8_t* s_listed;
  const uint8_t* s_changed;
  const uint8_t* s_vis_changed;
  const int32_t* s_turn;
  const uint8_t* s_done;
  const int32_t* s_width;
  const int32_t* s_height;
  const int32_t* s_players;
  const uint8_t* s_alive;
  const int32_t* s_army_count;
  const int32_t* s_general_idx;
  int32_t stride, max_p, max_w, max_h, fd, row_dw;
  const uint32_t* zeros;  // see StepArgs::zeros (setup_kernel)
  int32_t dst_envs;  // envs in the destination arrays: env ids are checked against it on the device
  uint32_t fresh;  // 1: start from a blank engine (reset); 0: poke the resident record
  uint32_t init;   // 1: run performInitialSetup after the import
  uint32_t fog;
  int32_t* status;  // device word, set to a GVEC_E_* code on contract violation
};

struct ExportArgs {
  const uint32_t* hdr;
  const uint32_t* rows;
  const uint32_t* army16;
  const int32_t* army32;
  int32_t env_begin, n;
  int32_t* army_out;
  int8_t* owner;
  uint8_t* type;
  uint8_t* visible;
  int8_t* listed;
  uint8_t* changed;
  uint8_t* vis_changed;
  int32_t* turn;
  uint8_t* done;
  int8_t* winner;
  int32_t* width;
  int32_t* height;
  int32_t* players;
  uint8_t* alive;
  int32_t* army_count;
  int32_t* tile_count;
  int32_t* general_idx;
  // ComputePlayerVisibility outputs
  int32_t vis_player;
  uint8_t* pv_visible;
  uint8_t* pv_fog;
  int32_t stride, max_p, fd, row_dw;
};

// gvec_export_records / gvec_import_records: resident envs [env_begin, env_begin+n) <-> a record slab
struct RecordArgs {
  uint32_t* hdr;
  uint32_t* rows;
  uint32_t* army16;
  int32_t* army32;
  uint32_t* rec_hdr;   // [n][HDR_DW]
  uint32_t* rec_rows;  // [n][row_dw]
  int32_t* rec_army;   // [n][NSLOT*64], always int32
  int32_t env_begin, n, fd, row_dw, max_w, max_h, max_p;
  int32_t* status;
};

struct MapgenArgs {
  int32_t* army;   // [n][stride]
  int8_t* owner;
  uint8_t* type;
  int32_t* width;  // [n] out (what was generated)
  int32_t* height;
  int32_t* players;
  const int32_t* in_width;  // [n] or null = max
  const int32_t* in_height;
  const int32_t* in_players;
  int32_t n, stride, max_w, max_h, max_p, first_index;
  uint32_t seed_lo, seed_hi;
  int32_t* status;
  // Go-seeded generation (gvec_reset_go_seeded): non-null selects Go's math/rand, one seed per board;
  // go_state: [607][n] u64 of scratch for the generators' vectors
  const int64_t* go_seeds;
  uint64_t* go_state;
};

// one turn per launch (KF_AGENT selects the on-device agent; the legal buffer must then be current)
hipError_t launch_step(const Variant& v, const StepArgs& a, hipStream_t s);
// a.turns fused turns per launch with the on-device agent
hipError_t launch_rollout(const Variant& v, const StepArgs& a, hipStream_t s);
hipError_t launch_agent(const Variant& v, const StepArgs& a, hipStream_t s);
hipError_t launch_legal(const Variant& v, const StepArgs& a, hipStream_t s);
hipError_t launch_serializer_mask(const Variant& v, const StepArgs& a, hipStream_t s);

// internal/experience side channel
struct ExperienceArgs {
  const uint32_t* hdr;
  const uint32_t* rows;
  const uint32_t* army16;
  const int32_t* army32;
  uint32_t* snap;     // [B][snap_dw]: see SnapLayout (gvec_kernels.hip); indexed by env
  float* rewards;     // [n][pstride]
  uint8_t* done;      // [n] or null
  float* obs;         // observe: [B][9*stride] (one player) or [B][pstride][9*stride] (player = -1)
  uint32_t* records;  // experience records [n][record_dw] (RecordLayout)
  const gvec_action* actions;  // [B][pstride]: what was played in the step between snapshot and record
  int32_t env_begin;  // snapshot / rewards / records work on envs [env_begin, env_begin + num_envs)
  int32_t num_envs, fd, row_dw, snap_dw, record_dw, pstride, stride, player, env_id_base;
};
hipError_t launch_experience_records(const Variant& v, const ExperienceArgs& a, hipStream_t s);
// dwords of one env's snapshot / experience record for this variant
void experience_layout(const Variant& v, int fd, int* snap_dw, int* record_dw);
// python/generals_gym on the device
struct GymArgs {
  const uint32_t* hdr;
  const uint32_t* rows;
  const uint32_t* army16;
  const int32_t* army32;
  const int64_t* turn_count;  // [B] GeneralsEnv.turn_count
  float* obs;                 // [B][9][stride]
  uint8_t* mask;              // [B][stride*5]
  double* reward;             // [B] or null
  uint8_t* done;              // [B] or null
  int8_t* winner;             // [B] or null
  int32_t* prev_stats;        // [B][3*MAXP]: tile_count, army_count, alive as of the previous call (read, then rewritten)
  // gvec_gym_finish_step only (played == null: plain gvec_gym_observe): GeneralsEnv.step's bookkeeping, per env
  const uint8_t* resetting;   // [B] in: this step re-dealt the env
  const uint8_t* played;      // [B] in: the action was submitted (gvec_gym_actions)
  int64_t* turn_io;           // [B] in / out: turn_count (the same array as turn_count)
  int64_t* turn_out;          // [B] out: a copy for the caller's info dict
  uint8_t* terminated;        // [B] out 0/1
  uint8_t* truncated;         // [B] out 0/1
  uint8_t* needs_reset;       // [B] out 0/1: terminated | truncated
  int32_t num_envs, fd, row_dw, stride, player, max_turns;
};
struct GymActArgs {
  const uint32_t* hdr;
  const int64_t* gym_actions;  // [B] indices into Discrete(board_size * 5)
  const uint8_t* mask;         // [B][stride*5] of the last gym_observe
  const uint8_t* resetting;    // [B] or null
  gvec_action* actions;        // [B][pstride] in / out
  uint8_t* played;             // [B] outputs, any may be null
  uint8_t* invalid;
  uint8_t* error;
  int32_t num_envs, stride, pstride, player;
};
// gvec_gym_step: GeneralsEnv.step for every env in one launch (gym_step_kernel)
struct GymStepArgs {
  const int64_t* gym_actions;  // [B] indices into Discrete(board_size * 5)
  const uint8_t* resetting;    // [B] needs_reset of the previous step
  int64_t* turn_io;            // [B] in / out: GeneralsEnv.turn_count
  int64_t* turn_out;           // [B] out or null
  float* obs;                  // [B][9][stride]
  uint8_t* mask;               // [B][stride*5]
  double* reward;              // [B] outputs, any may be null
  uint8_t* terminated;
  uint8_t* truncated;
  int8_t* winner;
  uint8_t* needs_reset;
  uint8_t* played;
  uint8_t* invalid;
  uint8_t* error;
  int32_t* prev_stats;         // [B][3*MAXP], as GymArgs
  int32_t stride, player, max_turns;
};
// gvec_stream_deltas: createStreamUpdate's delta for one player's stream (server.go:636-777)
struct StreamDeltaArgs {
  const uint32_t* hdr;
  const uint32_t* rows;
  const uint32_t* army16;
  const int32_t* army32;
  uint8_t* kind;                 // [B] 1 delta, 2 the server would send the full state
  int32_t* count;                // [B] tile updates of the delta
  unsigned long long* updates;   // [B][cap]
  int32_t num_envs, fd, row_dw, player, cap;
  int32_t full_tiles;            // != 0: an env of kind 2 gets ALL its tiles (count = W*H, cap >= tile stride), the full state's board
};
hipError_t launch_stream_deltas(const Variant& v, const StreamDeltaArgs& a, hipStream_t s);
// offset[n + 1] = exclusive prefix sum of count[n]; packed[offset[e] + k] = rows[e][k] for k < count[e] (entries beyond `capacity` dropped)
hipError_t launch_pack_updates(const unsigned long long* rows, const int32_t* count, long long* offset, unsigned long long* packed, int32_t n,
                               int32_t cap, long long capacity, hipStream_t s);
// the consumer side of the record exchange: layout8 as gvec_experience_record_layout fills it
hipError_t launch_expand_records(const void* records, int32_t n, const int32_t* layout8, float* state, float* next_state, uint8_t* mask,
                                 int32_t* meta, hipStream_t s);
// gvec_pool_collect: ParallelEnvPool._run_episode's bookkeeping + ReplayBuffer.push for every worker (generals_vec.h)
hipError_t launch_pool_collect(const gvec_collect_args& a, hipStream_t s);
size_t pool_collect_scratch_bytes(int32_t num_envs);
hipError_t launch_gym_step(const Variant& v, const StepArgs& a, const GymStepArgs& g, hipStream_t s);
hipError_t launch_gym_observe(const Variant& v, const GymArgs& a, hipStream_t s);
hipError_t launch_gym_actions(const GymActArgs& a, hipStream_t s);
hipError_t launch_snapshot(const Variant& v, const ExperienceArgs& a, hipStream_t s);
hipError_t launch_rewards(const Variant& v, const ExperienceArgs& a, hipStream_t s);
hipError_t launch_observe(const Variant& v, const ExperienceArgs& a, hipStream_t s);
hipError_t launch_import(const Variant& v, const ImportArgs& a, hipStream_t s);
// performInitialSetup for the envs the import of the same ImportArgs marked (a.init)
hipError_t launch_setup(const Variant& v, const ImportArgs& a, hipStream_t s);
hipError_t launch_export(const Variant& v, const ExportArgs& a, hipStream_t s);
hipError_t launch_records(const Variant& v, const RecordArgs& a, bool import, hipStream_t s);
hipError_t launch_mapgen(const MapgenArgs& a, hipStream_t s);
// sums the H_CNT_* counters of all envs into out[3] (u64, device)
hipError_t launch_counter_sum(const uint32_t* hdr, int32_t num_envs, unsigned long long* out, hipStream_t s);
// device self-test of the wave primitives; out[0] = 0 on success else a failing check id
hipError_t launch_selftest(int32_t* out, hipStream_t s);

}  // namespace gvec
