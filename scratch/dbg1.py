import sys, os
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
import _harness as H, _oracle as O
import generalsreinforcementlearning_amd as g
B=512
sizes=[(10,10,2)]*B
army, owner, typ, w, h, p = H.gen_boards(1234, sizes, 10, 10)
eng = g.VecEngine(B,10,10,2, fog_of_war=False)
eng.reset(army, owner, typ, w, h, p)
st = eng.game_state()
bad = np.argwhere(st["type"] != typ)
print("after reset: type mismatches", len(bad), bad[:10])
t0=(np.arange(B)%25).astype(np.int32)
eng.write_state({"turn": t0})
st = eng.game_state()
bad = np.argwhere(st["type"] != typ)
print("after write_state: type mismatches", len(bad))
for e,t in bad[:20]:
    print(e, t, "x,y=", t%10, t//10, "hip", st["type"][e,t], "exp", typ[e,t], "turn", st["turn"][e])
import collections
print(collections.Counter((int(t)//10) for e,t in bad))
print(collections.Counter((int(e)%4) for e,t in bad))
print("other fields:", {f: int((st[f]!=v).sum()) for f,v in (("army",army),("owner",owner))})
