#!/bin/bash
# Builds profiling variants of the library with one phase of the turn compiled out each (GVEC_PROFILE_SKIP bits:
# 1 agent, 2 fog, 4 action phase, 8 production, 16 end-of-turn stats, 32 gt1 refresh, 64 legal-mask emission) into ab/prof_*.so.
# Run in the build container; then on the GPU box: scripts/pmc_insts.sh ab/prof_*.so
cd "$(dirname "$0")/../generalsreinforcementlearning_amd/csrc"
mkdir -p ../../ab
for bits in 0 1 2 4 8 16 32 64; do
  ( hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DGVEC_PROFILE_SKIP=$bits -shared -o ../../ab/prof_$bits.so gvec_kernels.hip gvec_api.hip 2>&1 | grep -i error ) &
done
wait
# ... and with one idempotent phase run TWICE each (GVEC_PROFILE_DUP, same bits): the boards play the same games as the
# plain build, so the counter difference is the phase's dynamic cost
for bits in 1 2 16 32 64; do
  ( hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DGVEC_PROFILE_DUP=$bits -shared -o ../../ab/dup_$bits.so gvec_kernels.hip gvec_api.hip 2>&1 | grep -i error ) &
done
wait
ls -la ../../ab
