#!/usr/bin/env python3
"""Register / scratch / LDS use of every kernel variant matching a pattern (compiles gvec_kernels.hip with -save-temps).
usage: scripts/kernel_regs.py [substring, default Li4ELi7]"""
import os, re, subprocess, sys
pat = sys.argv[1] if len(sys.argv) > 1 else "Li4ELi7"
src = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "generalsreinforcementlearning_amd", "csrc", "gvec_kernels.hip")
os.makedirs("/tmp/gvec_regs", exist_ok=True)
subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-c", src, "-o", "k.o", "-save-temps"],
               cwd="/tmp/gvec_regs", check=True, capture_output=True)
s = open("/tmp/gvec_regs/gvec_kernels-hip-amdgcn-amd-amdhsa-gfx950.s").read()
for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", s, re.S):
    if pat not in m.group(1):
        continue
    g = lambda k: re.search(r"\.amdhsa_" + k + r"\s+(\S+)", m.group(2)).group(1)
    print(f"{m.group(1)[:64]:64s} vgpr {g('next_free_vgpr'):>4s} sgpr {g('next_free_sgpr'):>4s} scratch {g('private_segment_fixed_size'):>5s} lds {g('group_segment_fixed_size'):>5s}")
