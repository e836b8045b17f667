import sys, os, ctypes as C
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
import _harness as H, _oracle as O
import generalsreinforcementlearning_amd as g
hip = C.CDLL("libamdhip64.so")
B=512
sizes=[(10,10,2)]*B
army, owner, typ, w, h, p = H.gen_boards(1234, sizes, 10, 10)
eng = g.VecEngine(B,10,10,2, fog_of_war=False)
eng.reset(army, owner, typ, w, h, p)
eng.synchronize()
hs, M = 12, 11
rows = np.zeros((B, M, hs), np.uint32)
ptr = eng.device_buffer(1)
rc = hip.hipMemcpy(C.c_void_p(rows.ctypes.data), C.c_void_p(ptr), C.c_size_t(rows.nbytes), 2)
print("memcpy rc", rc)
def plane_from(mask2d):  # [B,100] bool -> [B,hs] rows
    out = np.zeros((B, hs), np.uint32)
    m = mask2d.reshape(B,10,10)
    for y in range(10):
        for x in range(10):
            out[:, y] |= (m[:, y, x].astype(np.uint32) << x)
    return out
exp = {8: plane_from(typ==1), 9: plane_from(typ==2), 10: plane_from(typ==3), 0: plane_from(owner==0), 1: plane_from(owner==1)}
for m, e in exp.items():
    bad = np.argwhere(rows[:, m, :] != e)
    print("plane", m, "mismatching rows:", len(bad), bad[:6].tolist())
    for (env, y) in bad[:4]:
        print("   env", env, "y", y, "got", bin(rows[env, m, y]), "exp", bin(e[env, y]))
st = eng.game_state()
print("export type mismatches:", int((st["type"] != typ).sum()))
