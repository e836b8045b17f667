"""TEST INFRASTRUCTURE: the experience-record format of include/generals_vec.h written down a second time,
from the ORACLE's state (plain numpy, one record at a time).  The GPU test compares the HIP kernel's records with
these byte for byte; the gloo test ships these over the N>1 gather path and decodes them with the product decoder."""
import numpy as np

import _oracle as O


def layout_for(max_w, max_h, max_p):
    """What gvec_experience_record_layout reports for a handle of these limits (gvec_kernels.hip pick_variant,
    gvec_api.hip plane_dwords, RecordLayout)."""
    stride = max_w * max_h
    mp = next(m for m in (2, 4, 8) if m >= max_p)
    ns = next(n for n in (1, 2, 4, 7, 10, 16) if n * 64 >= stride)
    fd = 2 * ns - 1 if stride <= 32 * (2 * ns - 1) else 2 * ns
    army_next = 4 + 2 * mp + (4 * mp + 3) * fd + 4 * mp * fd + ns * 32
    return {"record_dw": (army_next + ns * 32 + 3) // 4 * 4, "mp": mp, "fd": fd, "ns": ns, "max_players": max_p, "stride": stride}


def _plane(bits, fd):
    """bool[N] -> uint32[fd] (bit t of the string = bit t & 31 of dword t >> 5)."""
    b = np.zeros(fd * 32, np.uint8)
    b[: len(bits)] = bits
    return np.packbits(b, bitorder="little").view(np.uint32)


def capture(ora):
    """What gvec_experience_begin snapshots: call BEFORE the step (and before ora.experience_begin())."""
    st = ora.read_state()
    return {"st": st, "mask": ora.serializer_mask()}


def encode(ora, snap, acts, layout, env_id_base=0, envs=None):
    """The records of `envs` (default: all) after the step; needs ora.experience_begin() before / this after the step."""
    from generalsreinforcementlearning_amd.experience import record_offsets
    off, mp, fd, ns, rd = record_offsets(layout), layout["mp"], layout["fd"], layout["ns"], layout["record_dw"]
    cur = ora.read_state()
    rewards, done = ora.rewards()
    prev = snap["st"]
    envs = range(ora.B) if envs is None else envs
    out = np.zeros((len(envs), rd), np.uint32)
    for i, e in enumerate(envs):
        w, h, P = int(cur["width"][e]), int(cur["height"][e]), int(cur["players"][e])
        n = w * h
        r = out[i]
        comparable = (prev["width"][e] == w and prev["height"][e] == h and cur["turn"][e] > prev["turn"][e])
        fog = bool(ora._prm.fog_of_war)
        r[0] = np.uint32(np.int32(cur["turn"][e]))
        r[1] = w | (h << 8) | (P << 16) | (((1 if done[e] else 0) | (2 if fog else 0) | (4 if comparable else 0)) << 24)
        acted = 0
        pw = int(prev["width"][e])
        for p in range(P):
            a = acts[e, p]
            if not (a["flags"] & 1):
                r[off["action"] + p] = np.uint32(0xFFFFFFFF)
                continue
            acted |= 1 << p
            dx, dy = int(a["to_x"]) - int(a["from_x"]), int(a["to_y"]) - int(a["from_y"])
            d = {(0, 1): 1, (-1, 0): 2, (1, 0): 3}.get((dx, dy), 0)
            r[off["action"] + p] = np.uint32(np.int32((int(a["from_y"]) * pw + int(a["from_x"])) * 4 + d))
        for p in range(P, mp):
            r[off["action"] + p] = np.uint32(0xFFFFFFFF)
        r[2] = acted
        r[3] = env_id_base + e
        r[off["reward"]: off["reward"] + ora.max_p] = rewards[e].view(np.uint32)
        pl = r[off["planes"]: off["planes"] + (4 * mp + 3) * fd].reshape(4 * mp + 3, fd)
        pn = int(prev["width"][e]) * int(prev["height"][e])
        for p in range(mp):
            pl[p] = _plane(prev["owner"][e, :pn] == p, fd)
            pl[mp + p] = _plane((prev["visible"][e, :pn] >> p) & 1, fd)
            pl[2 * mp + p] = _plane(cur["owner"][e, :n] == p, fd)
            pl[3 * mp + p] = _plane((cur["visible"][e, :n] >> p) & 1, fd)
        pl[4 * mp + 0] = _plane(cur["type"][e, :n] == 1, fd)
        pl[4 * mp + 1] = _plane(cur["type"][e, :n] == 2, fd)
        pl[4 * mp + 2] = _plane(cur["type"][e, :n] == 3, fd)
        m = r[off["mask"]: off["mask"] + 4 * mp * fd].reshape(mp, 4 * fd)
        m[: ora.max_p] = snap["mask"][e].view(np.uint32).reshape(ora.max_p, 4 * fd)
        for name, src in (("army_prev", prev["army"][e]), ("army_next", cur["army"][e])):
            a16 = r[off[name]: off[name] + ns * 32].view(np.uint16)
            a16[: len(src)] = np.clip(src, 0, 65535).astype(np.uint16)
    return out
