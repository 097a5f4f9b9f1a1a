"""Independent pure-Python restatement of the hot path, written from src/phylomap.cpp (not from the C oracle)
to pin oracle/phm_oracle.c on small cases: same arithmetic spec (IEEE doubles, left-to-right unfused sums,
Philox4x32-7 streams, phm_log/phm_exp), so the two must agree BIT FOR BIT.  Small trees only (plain loops).

TEST INFRASTRUCTURE ONLY.
"""
import math
import struct
from fractions import Fraction

M32 = 0xFFFFFFFF
ENT_NODE, ENT_BSTATE, ENT_BEXP, ENT_BUNIF = 0, 1 << 30, 2 << 30, 3 << 30


STREAM_ROUNDS = 7       # Philox4x32-7: the sampler's streams (the fewest rounds that pass BigCrush; Random123's default is 10)


def philox(ctr, key, rounds=STREAM_ROUNDS):
    c0, c1, c2, c3 = ctr
    k0, k1 = key
    for _ in range(rounds):
        p0 = 0xD2511F53 * c0
        p1 = 0xCD9E8D57 * c2
        c0, c1, c2, c3 = ((p1 >> 32) ^ c1 ^ k0) & M32, p1 & M32, ((p0 >> 32) ^ c3 ^ k1) & M32, p0 & M32
        k0 = (k0 + 0x9E3779B9) & M32
        k1 = (k1 + 0xBB67AE85) & M32
    return c0, c1, c2, c3


def u01(x):
    return (float(x) + 0.5) * 2.0 ** -32


def d2u(x):
    return struct.unpack("<Q", struct.pack("<d", x))[0]


def u2d(u):
    return struct.unpack("<d", struct.pack("<Q", u))[0]


def plog(x):
    ln2_hi, ln2_lo = 6.93147180369123816490e-01, 1.90821492927058770002e-10
    Lg = [6.666666666666735130e-01, 3.999999999940941908e-01, 2.857142874366239149e-01, 2.222219843214978396e-01,
          1.818357216161805012e-01, 1.531383769920937332e-01, 1.479819860511658591e-01]
    k = 0
    ux = d2u(x)
    hx = ux >> 32
    k += (hx >> 20) - 1023
    hx &= 0xFFFFF
    i = (hx + 0x95F64) & 0x100000
    ux = ((hx | (i ^ 0x3FF00000)) << 32) | (ux & M32)
    k += i >> 20
    f = u2d(ux) - 1.0
    dk = float(k)
    s = f / (2.0 + f)
    z = s * s
    w = z * z
    t1 = w * (Lg[1] + w * (Lg[3] + w * Lg[5]))
    t2 = z * (Lg[0] + w * (Lg[2] + w * (Lg[4] + w * Lg[6])))
    R = t2 + t1
    hfsq = 0.5 * f * f
    return dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f)


def pexp(x):
    ln2HI, ln2LO, invln2 = 6.93147180369123816490e-01, 1.90821492927058770002e-10, 1.44269504088896338700e+00
    P = [1.66666666666666019037e-01, -2.77777777770155933842e-03, 6.61375632143793436117e-05,
         -1.65339022054652515390e-06, 4.13813679705723846039e-08]
    if x > 7.09782712893383973096e+02:
        return math.inf
    if x < -7.45133219101941108420e+02:
        return 0.0
    hi, lo, k = x, 0.0, 0
    if abs(x) > 0.34657359027997264:
        k = int(invln2 * x + (-0.5 if x < 0.0 else 0.5))
        t = float(k)
        hi = x - t * ln2HI
        lo = t * ln2LO
    r = hi - lo
    t = r * r
    c = r - t * (P[0] + t * (P[1] + t * (P[2] + t * (P[3] + t * P[4]))))
    y = 1.0 - ((lo - (r * c) / (2.0 - c)) - hi)
    return math.ldexp(y, k)


def _logtab():
    import importlib.util
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "gen_log_table.py")
    spec = importlib.util.spec_from_file_location("gen_log_table", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.table()


_INV, _LOGC = _logtab()


def neglog_u32(k):
    """-log((k + 0.5) 2^-32): the engine's exponential variate (phm_device.h neglog_u32, oracle orc_neglog_u32)."""
    y = float(k) * 2.0 + 1.0
    f, e = math.frexp(y)
    if e == 33 and f >= 0.99609375:
        r, c0, ee = f - 1.0, 0.0, 0.0
    else:
        j = int((f - 0.5) * 256.0)
        c = 0.501953125 + float(j) * 0.00390625
        r, c0, ee = (f - c) * _INV[j], _LOGC[j], float(e - 33)
    p = 1.0 / 7.0
    p = p * r - 1.0 / 6.0
    p = p * r + 0.2
    p = p * r - 0.25
    p = p * r + 1.0 / 3.0
    p = p * r - 0.5
    p = p * r * r + r
    return -(ee * 6.93147180369123816490e-01 + (c0 + (p + ee * 1.90821492927058770002e-10)))


class Rng:
    def __init__(self, seed, replica):
        self.key = (seed & M32, (seed >> 32) & M32)
        self.rep = replica

    def u(self, it, ent, d):
        o = philox((d >> 2, ent, it, self.rep), self.key)
        return u01(o[d & 3])

    def e(self, it, ent, d):
        o = philox((d >> 2, ent, it, self.rep), self.key)
        return neglog_u32(o[d & 3])


def fma(a, b, c):
    """round(a*b + c) with ONE rounding: exact rational arithmetic, then int/int true division, which CPython rounds
    correctly (ties to even).  All operands here are finite and non-negative."""
    if a == 0.0 or b == 0.0:
        return c
    x = Fraction(a) * Fraction(b) + Fraction(c)
    return x.numerator / x.denominator


def matvec_lr(M, v):
    """unfused left-to-right sums for every n: the EXP path (makePLexp :2899-2906, newunifSample :127)"""
    n = len(v)
    out = []
    for i in range(n):
        acc = M[i][0] * v[0]
        for j in range(1, n):
            acc += M[i][j] * v[j]
        out.append(acc)
    return out


def matvec(M, v):
    """The MCMC sweep's chains.  n <= 4: unfused left-to-right sums (Armadillo gemv_emul_tinysq).  n > 4: the reference's order is BLAS dgemv
    (unknowable), so the spec is what the MI355X matrix cores compute: a fused multiply-add per term, j ascending, from +0."""
    n = len(v)
    out = []
    for i in range(n):
        if n > 4:
            acc = 0.0
            for j in range(n):
                acc = fma(M[i][j], v[j], acc)
        else:
            acc = M[i][0] * v[0]
            for j in range(1, n):
                acc += M[i][j] * v[j]
        out.append(acc)
    return out


def matTvec(M, v):
    n = len(v)
    out = []
    for c in range(n):
        if n > 4:
            acc = 0.0
            for r in range(n):
                acc = fma(M[r][c], v[r], acc)
        else:
            acc = M[0][c] * v[0]
            for r in range(1, n):
                acc += M[r][c] * v[r]
        out.append(acc)
    return out


def rowsum(p):
    """normalisation sum (:525).  n > 4: four interleaved partial sums (states g, g+4, ...), then (t0+t1)+(t2+t3)."""
    n = len(p)
    if n <= 4:
        s = p[0]
        for c in range(1, n):
            s += p[c]
        return s
    t = [0.0, 0.0, 0.0, 0.0]
    for c in range(n):
        t[c & 3] += p[c]
    return (t[0] + t[1]) + (t[2] + t[3])


def sample(p, u):
    total = p[0]
    for x in p[1:]:
        total += x
    assert total > 0.0
    thr = u * total
    cum = p[0]
    if thr <= cum:
        return 0
    for j in range(1, len(p)):
        cum += p[j]
        if thr <= cum:
            return j
    raise AssertionError("unreachable with a matching summation order")


def sumstatMCMC(z, Q, pid, Omega, N, nen, nodelist, root, seed, replica, variant="plain"):
    """maketreelistMCMC / _bigtree / SPARSE (src/phylomap.cpp:891-986, 822-870) with treesample (:775-785)."""
    n = len(Q)
    E = len(z["edge"])
    T = len(z["states"])
    e1 = [int(r[0]) for r in z["edge"]]
    e2 = [int(r[1]) for r in z["edge"]]
    B2 = [[(1.0 if i == j else 0.0) + Q[i][j] / Omega for j in range(n)] for i in range(n)]
    Bc = [[(b if b > 1e-7 else 0.0) for b in row] for row in B2] if variant == "sparse" else B2    # :811
    rng = Rng(seed, replica)
    dw = [[float(x) for x in z["maps"][b]] for b in range(E)]
    st = [[int(x) - 1 for x in z["mapnames"][b]] for b in range(E)]                              # :29
    PL = [[0.0] * n for _ in range(2 * T - 1)]
    hid = variant == "ks"                                  # treesampleks :1422-1432 with Q fixed (maketreelistMCMCks :1802-1872)
    ks = hid or variant == "bf"                            # bf: treesamplebf :1169-1179 -- the n x n counting layout with observed tips
    kk = n // 2 - 1 if hid else 0
    for i in range(T):
        if not hid:
            PL[i][int(z["states"][i]) - 1] = 1.0                                                # :914
        else:                                                                                   # :1838-1845
            for j in range(1 if int(z["states"][i]) % 2 == 0 else 0, n, 2):
                PL[i][j] = 1.0
    cols = n + n * n + 2 + 3 * kk + 1 if ks else n + n * (n - 1)
    out = [[0.0] * cols for _ in range(N)]
    for it in range(N):
        if ks:                                                                                  # recordQks :1789-1798
            base = n + n * n
            out[it][base], out[it][base + 1] = Q[0][1], Q[1][0]
            for i in range(kk):
                out[it][base + 2 + i] = Q[2 * i][2 * i + 2]
                out[it][base + 2 + kk + i] = Q[2 * i + 2][2 * i]
                out[it][base + 2 + 2 * kk + i] = Q[2 * (i + 1)][2 * (i + 1) + 1] / Q[0][1]
        m = [len(d) for d in dw]                                                                # :598-599
        for i in range(T - 1):                                                                  # makePLrcpp :503-514
            ea, eb = nen[2 * i] - 1, nen[2 * i + 1] - 1
            first = list(PL[e2[eb] - 1])
            second = list(PL[e2[ea] - 1])
            for _ in range(m[eb] - 1):
                first = matvec(Bc, first)
            for _ in range(m[ea] - 1):
                second = matvec(Bc, second)
            row = [first[c] * second[c] for c in range(n)]
            if variant == "bigtree" or ks:                                                      # :525 / :1085
                s = rowsum(row)
                row = [x / s for x in row]
            PL[e1[ea] - 1] = row
        rm = [0] * (2 * T - 1)
        for i in range(T):
            rm[i] = int(z["states"][i]) - 1
        rm[root - 1] = sample([pid[c] * PL[root - 1][c] for c in range(n)], rng.u(it, ENT_NODE | (root - 1), 0))   # :618-627
        for node in nodelist:                                                                   # :640-657
            j = e2.index(node)
            ps = rm[e1[j] - 1]
            v = [0.0] * n
            v[ps] = 1.0
            for _ in range(m[j] - 1):
                v = matTvec(Bc, v)
            rm[node - 1] = sample([v[c] * PL[node - 1][c] for c in range(n)], rng.u(it, ENT_NODE | (node - 1), 0))
        if ks:
            out[it][cols - 1] = float(rm[root - 1])                                             # :1350-1352 / :1129
        if hid:
            for b in range(E):                                                                  # :1384-1397
                if e2[b] <= T:
                    ps = rm[e1[b] - 1]
                    v = [0.0] * n
                    v[ps] = 1.0
                    for _ in range(m[b] - 1):
                        v = matTvec(Bc, v)
                    rm[e2[b] - 1] = sample([v[c] * PL[e2[b] - 1][c] for c in range(n)], rng.u(it, ENT_NODE | (e2[b] - 1), 0))
        for b in range(E):                                                                      # updatenodestates :460-475
            st[b][0] = rm[e1[b] - 1]
            st[b][-1] = rm[e2[b] - 1]
        for b in range(E):                                                                      # sampleabranch :370-413
            ss = len(dw[b])
            if ss > 2:                                                                          # resamplebranchstates :264-308
                beta = [[0.0] * n]
                beta[0][st[b][-1]] = 1.0
                for j in range(1, ss - 1):
                    beta.append(matvec(Bc, beta[j - 1]))
                for i in range(1, ss - 1):
                    p = [B2[st[b][i - 1]][c] * beta[ss - i - 1][c] for c in range(n)]
                    st[b][i] = sample(p, rng.u(it, ENT_BSTATE | b, i - 1))
            if ks:                                                                              # shortenerbf :1010-1014
                for i in range(1, ss):
                    out[it][n + st[b][i - 1] * n + st[b][i]] += 1.0
            nd, ns = [dw[b][0]], [st[b][0]]                                                     # shortener :44-73
            for i in range(1, ss):
                if st[b][i] != ns[-1]:
                    nd.append(dw[b][i]); ns.append(st[b][i])
                else:
                    nd[-1] = nd[-1] + dw[b][i]
            for i in range(1, len(ns)):
                a, c = ns[i - 1], ns[i]
                if not ks:
                    out[it][n + a * (n - 1) + (c - 1 if a < c else c)] += 1.0
            fd, fs, ed = [], [], 0                                                              # virtual jumps :391-410
            for seglen, s in zip(nd, ns):
                scale = 1.0 / (Omega + Q[s][s])
                tot = 0.0
                while tot < seglen:
                    rl = scale * rng.e(it, ENT_BEXP | b, ed)
                    ed += 1
                    if tot + rl < seglen:
                        fd.append(rl); fs.append(s); tot += rl
                    else:
                        fd.append(seglen - tot); fs.append(s); tot = seglen
            dw[b], st[b] = fd, fs
        for b in range(E):                                                                      # updatedwelltimes :745-757
            for d, s in zip(dw[b], st[b]):
                out[it][s] += d
    return out


def matexp(L, R, dv, t):
    """matexp :2964-2968 then abs (:2980)."""
    n = len(dv)
    e = [pexp(dv[k] * t) for k in range(n)]
    P = [[0.0] * n for _ in range(n)]
    for i in range(n):
        for j in range(n):
            acc = (L[i][0] * e[0]) * R[0][j]
            for k in range(1, n):
                acc += (L[i][k] * e[k]) * R[k][j]
            P[i][j] = abs(acc)
    return P


def sumstatEXP(z, Q, pid, N, nen, nodelist, root, L, R, dv, seed, replica, rescale=False):
    """maketreelistEXP :3001-3051 with treesampleEXP :2977-2996 and newunifSample :93-208."""
    n = len(Q)
    E = len(z["edge"])
    T = len(z["states"])
    e1 = [int(r[0]) for r in z["edge"]]
    e2 = [int(r[1]) for r in z["edge"]]
    tl = [float(x) for x in z["edge.length"]]
    rate = -1.0 * min(Q[i][i] for i in range(n))
    B2 = [[(1.0 if i == j else 0.0) + Q[i][j] / rate for j in range(n)] for i in range(n)]
    rng = Rng(seed, replica)
    P = [matexp(L, R, dv, tl[b]) for b in range(E)]
    PL = [[0.0] * n for _ in range(2 * T - 1)]
    for i in range(T):
        PL[i][int(z["states"][i]) - 1] = 1.0
    for i in range(T - 1):                                                                      # makePLold :2891-2893
        ea, eb = nen[2 * i] - 1, nen[2 * i + 1] - 1
        a = matvec_lr(P[ea], PL[e2[ea] - 1])
        b = matvec_lr(P[eb], PL[e2[eb] - 1])
        row = [a[c] * b[c] for c in range(n)]
        if rescale:                                   # not in the reference: row / sum(row), left to right (sampler-equivalent)
            sm = row[0]
            for c in range(1, n):
                sm += row[c]
            row = [x / sm for x in row]
        PL[e1[ea] - 1] = row
    cols = n + n * (n - 1)
    out = [[0.0] * cols for _ in range(N)]
    for it in range(N):
        rm = [0] * (2 * T - 1)
        for i in range(T):
            rm[i] = int(z["states"][i]) - 1
        rm[root - 1] = sample([pid[c] * PL[root - 1][c] for c in range(n)], rng.u(it, ENT_NODE | (root - 1), 0))
        for node in nodelist:
            j = e2.index(node)
            ps = rm[e1[j] - 1]
            rm[node - 1] = sample([P[j][ps][c] * PL[node - 1][c] for c in range(n)], rng.u(it, ENT_NODE | (node - 1), 0))
        for b in range(E):
            a, e = rm[e1[b] - 1], rm[e2[b] - 1]
            t, tp = tl[b], P[b][a][e]
            dr = 0
            rU = rng.u(it, ENT_BUNIF | b, dr); dr += 1
            lam = rate * t
            pk = pexp(-lam)
            cum = pk / tp if a == e else 0.0
            beta = [[0.0] * n]
            beta[0][e] = 1.0
            k = 0
            while not cum > rU:
                k += 1
                assert k <= 300
                beta.append(matvec_lr(B2, beta[k - 1]))
                pk = pk * lam / float(k)
                cum += pk * beta[k][a] / tp
            if k == 0 or (k == 1 and a == e):
                segs = [(a, t - 0.0)]
            elif k == 1:
                tj = t * rng.u(it, ENT_BUNIF | b, dr); dr += 1
                segs = [(a, tj - 0.0), (e, t - tj)]
            else:
                times = []
                for _ in range(k):
                    times.append(t * rng.u(it, ENT_BUNIF | b, dr)); dr += 1
                times.sort()
                dom = [a] + [0] * (k - 1) + [e]
                for i in range(1, k):
                    w = [B2[dom[i - 1]][c] * beta[k - i][c] for c in range(n)]
                    total = w[0]
                    for x in w[1:]:
                        total += x
                    u = rng.u(it, ENT_BUNIF | b, dr); dr += 1
                    cumw, pick = 0.0, None
                    for c in range(n):                                                          # sampleOnce :81-90
                        cumw += w[c] / total
                        if u < cumw:
                            pick = c
                            break
                    assert pick is not None
                    dom[i] = pick
                segs, tprev, sprev = [], 0.0, a
                for i in range(1, k + 1):
                    if dom[i - 1] != dom[i]:
                        segs.append((sprev, times[i - 1] - tprev))
                        tprev, sprev = times[i - 1], dom[i]
                segs.append((sprev, t - tprev))
            for i in range(1, len(segs)):
                x, y = segs[i - 1][0], segs[i][0]
                out[it][n + x * (n - 1) + (y - 1 if x < y else y)] += 1.0
            for s, d in segs:
                out[it][s] += d
    return out
