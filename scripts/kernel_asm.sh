#!/bin/bash
# Dumps the ISA of one kernel (default: the headline step kernel) to /tmp/gvec_asm/k.s, comments stripped.
K=${1:-_ZN4gvec11step_kernelILi4ELi7ELb1ELb1EEEvNS_8StepArgsE}
mkdir -p /tmp/gvec_asm && cd /tmp/gvec_asm && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -c /root/repo/generalsreinforcementlearning_amd/csrc/gvec_kernels.hip -o k.o -save-temps 2>/dev/null
awk "/^$K:/,/\.Lfunc_end/" gvec_kernels-hip-amdgcn-amd-amdhsa-gfx950.s | grep -v "^\s*;\|\.loc\|Ltmp\|implicit-def" > k.s
wc -l k.s
