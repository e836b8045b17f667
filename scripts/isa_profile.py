#!/usr/bin/env python3
"""Static instruction mix of one kernel, bucketed by the source function each instruction was inlined from."""
import re, subprocess, sys, collections, os
kern = sys.argv[1] if len(sys.argv) > 1 else "_ZN4gvec11step_kernelILi4ELi7ELb1ELb1EEEvNS_8StepArgsE"
src = "/root/repo/generalsreinforcementlearning_amd/csrc/gvec_kernels.hip"
os.makedirs("/tmp/st2", exist_ok=True)
subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-gline-tables-only", "-c", src, "-o", "/tmp/st2/k.o", "-save-temps"], cwd="/tmp/st2", check=True, capture_output=True)
asm = open("/tmp/st2/gvec_kernels-hip-amdgcn-amd-amdhsa-gfx950.s").read().split("\n")
# file table
files = {}
for l in asm:
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
    if m: files[int(m.group(1))] = (m.group(3) or m.group(2))
# function line ranges from sources
def func_ranges(path):
    out = []; cur = None
    for i, l in enumerate(open(path), 1):
        m = re.match(r'\s*(?:template\s*<[^>]*>\s*)?(?:__device__|__global__).*?\b([A-Za-z_0-9]+)\s*\(', l)
        if m and not l.strip().startswith("//"): out.append((i, m.group(1)))
    return out
ranges = {}
for fid, f in files.items():
    p = f if os.path.isabs(f) else os.path.join("/root/repo/generalsreinforcementlearning_amd/csrc", f)
    if os.path.exists(p) and ("gvec" in p): ranges[fid] = func_ranges(p)
def which(fid, line):
    r = ranges.get(fid)
    if not r: return files.get(fid, "?").split("/")[-1]
    name = "?"
    for (l0, n) in r:
        if l0 <= line: name = n
        else: break
    return name
inside = False; cur = ("?", 0); counts = collections.Counter(); kinds = collections.defaultdict(collections.Counter)
for l in asm:
    if l.startswith(kern + ":"): inside = True; continue
    if inside and l.startswith(".Lfunc_end"): break
    if not inside: continue
    m = re.match(r'\s*\.loc\s+(\d+)\s+(\d+)', l)
    if m: cur = (int(m.group(1)), int(m.group(2))); continue
    s = l.strip()
    if not s or s.startswith((";", ".", "//")) or s.endswith(":"): continue
    op = s.split()[0]
    if not re.match(r'^[a-z]', op): continue
    fn = which(*cur)
    kind = "SALU" if op.startswith("s_") else "VALU" if op.startswith("v_") else "LDS" if op.startswith("ds_") else "VMEM" if op.startswith(("global_", "buffer_", "scratch_", "flat_")) else "other"
    counts[fn] += 1; kinds[fn][kind] += 1
tot = sum(counts.values())
print(f"{kern}: {tot} static instructions")
for fn, c in counts.most_common(40):
    k = kinds[fn]
    print(f"  {fn:22s} {c:6d}  VALU {k['VALU']:5d} SALU {k['SALU']:5d} LDS {k['LDS']:4d} VMEM {k['VMEM']:4d}")
