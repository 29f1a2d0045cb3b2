"""Offline LDS bank-conflict model of S3's (ocp_riccati_backward_reg_kernel) access patterns, with the banking rules of MI355X_MICROARCH.md:
ds_read_b64: two groups of 32 lanes, bank = (byte address / 4) mod 64, a lane touches two consecutive banks; a group takes 1 cycle + 1 per
extra distinct address on its busiest bank.  ds_write_b64: four groups of 16 contiguous lanes, bank = dword mod 32."""
import numpy as np
NV, NU, NX = 18, 12, 36
X_TRI = (NX * (NX + 1) // 2 + 1) // 2 * 2
K_QXX = 0; K_QXU = X_TRI; K_QUU = K_QXU + NX * NU; K_FQQ = K_QUU + NU * NU; K_FQV = K_FQQ + 36
K_FVQ = K_FQV + 36; K_FVV = K_FVQ + NV * NV; K_FVU = K_FVV + NV * NV; K_LX = K_FVU + NV * NU; K_LU = K_LX + NX; K_FX = K_LU + NU
ZERO_AT = K_FX + NX
def xsym(r, c): return c * (c + 1) // 2 + r if r <= c else r * (r + 1) // 2 + c
def nat(s): return 6 + s if s < 12 else s - 12 if s < 16 else 24 + (s - 16) if s < 28 else 18 + (s - 28) if s < 32 else 4 + (s - 32) if s < 34 else 22 + (s - 34) if s < 36 else s
DC = [0, 1, 1, 1, 1, 2]; DS = [3, 0, 1, 2, 3, 0]
TA = [2, 2, 2, 0, 0, 1]; TB = [0, 1, 2, 0, 1, 1]

def read_b64_cycles(addr_doubles):
    """addr_doubles[64]: double index per lane -> (cycles, ideal)"""
    cyc = 0
    for grp in (range(0, 32), range(32, 64)):
        per_bank = {}
        for l in grp:
            d = int(addr_doubles[l]) * 2
            for b in (d % 64, (d + 1) % 64):
                per_bank.setdefault(b, set()).add(int(addr_doubles[l]))
        cyc += max(len(v) for v in per_bank.values())
    return cyc, 2

def c_gather(base=0, perm=lambda off: off):
    out = []
    for d in range(6):
        for bb in range(3):
            addr = []
            for lane in range(64):
                g, li = lane >> 4, lane & 15
                rs = 16 * DC[d] + 4 * DS[d] + g
                rn = nat(rs); cn = nat(16 * bb + li)
                if rn < NV:
                    off = K_FQQ + rn + 6 * cn if cn < 6 else (K_FQV + rn + 6 * (cn - NV) if NV <= cn < NV + 6 else ZERO_AT)
                else:
                    off = K_FVQ + (rn - NV) + NV * cn
                addr.append(base + perm(off))
            out.append(("C d%d bb%d" % (d, bb), addr))
    return out

def q_gather(base=0, perm=lambda off: off):
    out = []
    for t in range(6):
        for q in range(4):
            addr = []
            for lane in range(64):
                g, li = lane >> 4, lane & 15
                ni, nj = nat(16 * TA[t] + 4 * q + g), nat(16 * TB[t] + li)
                if ni < NX and nj < NX: off = K_QXX + xsym(ni, nj)
                elif ni < NX: off = K_QXU + ni + NX * (nj - NX)
                elif nj < NX: off = K_QXU + nj + NX * (ni - NX)
                else: off = K_QUU + (ni - NX) + NU * (nj - NX)
                addr.append(base + perm(off))
            out.append(("Q t%d q%d" % (t, q), addr))
    return out

def mirror(LDP=49, PM=0):
    out = []
    for q in range(4):
        for name, f in (("P10", lambda g, li: PM + LDP * li + 16 + 4 * q + g),
                        ("m02", lambda g, li: PM + LDP * (32 + (li if li < 4 else 0)) + 4 * q + g),
                        ("m12", lambda g, li: PM + LDP * (32 + (li if li < 4 else 0)) + 16 + 4 * q + g),
                        ("d0", lambda g, li: PM + LDP * li + 4 * q + g),
                        ("d1", lambda g, li: PM + LDP * (16 + li) + 16 + 4 * q + g),
                        ("d2", lambda g, li: PM + LDP * (32 + li) + 32 + 4 * q + g)):
            out.append(("mirror %s q%d" % (name, q), [f(l >> 4, l & 15) for l in range(64)]))
    return out

def total(patterns):
    c = i = 0
    worst = []
    for name, addr in patterns:
        cy, idl = read_b64_cycles(addr)
        c += cy; i += idl
        worst.append((cy, name))
    return c, i, sorted(worst, reverse=True)

if __name__ == "__main__":
    for label, pats in (("C gather", c_gather()), ("Q gather", q_gather()), ("mirrorP (LDP 49)", mirror())):
        c, i, w = total(pats)
        print("%-20s %3d instructions  cycles %3d  ideal %3d  conflict %.0f %%   worst: %s" % (label, len(pats), c, i, 100.0 * (c - i) / c, w[:4]))
