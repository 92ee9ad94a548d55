"""Lane-level NumPy emulation of ts_ppo_q.h's data movement (v_mfma_f32_16x16x4_f32 operand / accumulator layouts, the
LDS tiles R1 / R2 / R3 / PP and the record tile with the kernel's index expressions, the slab layout and slab3_col_to_param) for one
32-sample tile of both networks, checked against plain matrix algebra.  A design / review tool (CPU only): it validates
the index arithmetic the kernel was written from, not the HIP source itself.

    python scripts/stepq_layout_emulator.py
"""
import numpy as np

rng = np.random.default_rng(0)
OBS, ACT, HID, K1S = 17, 6, 64, 5
K1 = 4 * K1S
PS, PF, ACT_PAD = 68, 40, 8
LANE = np.arange(64)
N_, G_ = LANE & 15, LANE >> 4


def mfma16(a, b, c):
    """a, b: [64] per-lane operands, c: [64, 4].  A[m = l & 15][k = l >> 4], B[k = l >> 4][n = l & 15],
    C/D: col = l & 15, row = 4 (l >> 4) + reg."""
    A = np.zeros((16, 4)); B = np.zeros((4, 16))
    A[N_, G_] = a
    B[G_, N_] = b
    D = A @ B
    out = c.copy()
    for r in range(4):
        out[:, r] += D[4 * G_ + r, N_]
    return out


def run(actor: bool):
    S = 32
    X = rng.normal(size=(S, OBS)); W1 = rng.normal(size=(HID, OBS)) * .3; b1 = rng.normal(size=HID) * .1
    W2 = rng.normal(size=(HID, HID)) * .2; b2 = rng.normal(size=HID) * .1
    NH = ACT if actor else 1
    WHd = rng.normal(size=(NH, HID)) * .3
    dout_true = rng.normal(size=(S, NH))                  # stands in for the loss derivative
    # ---- reference algebra
    Xaug = np.zeros((S, K1)); Xaug[:, :OBS] = X; Xaug[:, OBS] = 1
    W1aug = np.zeros((HID, K1)); W1aug[:, :OBS] = W1; W1aug[:, OBS] = b1
    H1 = np.tanh(Xaug @ W1aug.T); H2 = np.tanh(H1 @ W2.T + b2)
    head = H2 @ WHd.T
    dH2 = dout_true @ WHd; dZ2 = dH2 * (1 - H2 ** 2)
    dH1 = dZ2 @ W2; dZ1 = dH1 * (1 - H1 ** 2)
    dW2 = dZ2.T @ H1; db2 = dZ2.sum(0); dW1aug = dZ1.T @ Xaug; dWH = dout_true.T @ H2

    # ---- emulation: 4 waves, LDS tiles as flat arrays
    R1 = np.zeros(32 * PS); R2 = np.zeros(64 * PF); R3 = np.zeros(64 * PF)
    PP = np.zeros(4 * 2 * 8 * 16)
    REC_W = 28
    REC = rng.normal(size=32 * REC_W); REC.reshape(32, REC_W)[:, :OBS] = X      # act / adv / ... behind the observation: finite junk
    n, gq = N_, G_
    W = range(4)
    # resident weights per wave
    W1a = {w: [np.where(4 * j + gq < OBS, W1[16 * w + n, np.minimum(4 * j + gq, OBS - 1)], 0.0) for j in range(K1S)] for w in W}
    B1 = {w: np.stack([b1[16 * w + 4 * gq + r] for r in range(4)], 1) for w in W}
    W2f = {w: [W2[16 * w + n, 16 * (jr >> 2) + 4 * gq + (jr & 3)] for jr in range(16)] for w in W}
    W2t = {w: [W2[16 * (jr >> 2) + 4 * gq + (jr & 3), 16 * w + n] for jr in range(16)] for w in W}
    B2 = {w: np.stack([b2[16 * w + 4 * gq + r] for r in range(4)], 1) for w in W}
    if actor:
        WH = {w: [np.where(n < ACT, WHd[np.minimum(n, ACT - 1), 16 * w + 4 * gq + r], 0.0) for r in range(4)] for w in W}
        WHb = {w: [np.where(4 * r + gq < ACT, WHd[np.minimum(4 * r + gq, ACT - 1), 16 * w + n], 0.0) for r in range(2)] for w in W}
    else:
        WH = {w: [WHd[0, 16 * w + 4 * gq + r] for r in range(4)] for w in W}
    # phase 1
    for w in W:
        fb = 16 * w
        acc = [B1[w].copy(), B1[w].copy()]                   # bias = initial accumulator
        for j in range(K1S):
            k = 4 * j + gq
            kc = np.minimum(k, REC_W - 1)
            for b in range(2):
                acc[b] = mfma16(W1a[w][j], REC[(16 * b + n) * REC_W + kc], acc[b])     # fields >= obs meet zero weights
        for b in range(2):
            acc[b] = np.tanh(acc[b])
            for r in range(4):
                R1[(16 * b + n) * PS + fb + 4 * gq + r] = acc[b][:, r]
                R3[(fb + 4 * gq + r) * PF + 16 * b + n] = acc[b][:, r]
    assert np.allclose(R1.reshape(32, PS)[:, :64], H1)
    assert np.allclose(R3.reshape(64, PF)[:, :32], H1.T)
    # phase 2
    h2 = {}
    for w in W:
        fb = 16 * w
        acc = [B2[w].copy(), B2[w].copy()]
        for jj in range(4):
            for b in range(2):
                bv = np.stack([R1[(16 * b + n) * PS + 16 * jj + 4 * gq + r] for r in range(4)], 1)
                for r in range(4):
                    acc[b] = mfma16(W2f[w][4 * jj + r], bv[:, r], acc[b])
        h2[w] = [np.tanh(a) for a in acc]
        for b in range(2):
            for r in range(4):
                assert np.allclose(h2[w][b][:, r], H2[16 * b + n, fb + 4 * gq + r])
            if actor:
                for r in range(4):
                    R2[(fb + 4 * gq + r) * PF + 16 * b + n] = h2[w][b][:, r]
                pm = np.zeros((64, 4))
                for r in range(4):
                    pm = mfma16(WH[w][r], h2[w][b][:, r], pm)
                sel = gq < 2
                for r in range(4):
                    PP[(((w * 2 + b) * 8 + 4 * gq + r) * 16 + n)[sel]] = pm[sel, r]
            else:
                pv = sum(h2[w][b][:, r] * WH[w][r] for r in range(4))
                pv = pv + pv[LANE ^ 16]; pv = pv + pv[LANE ^ 32]
                sel = gq == 0
                PP[((w * 2 + b) * 16 + n)[sel]] = pv[sel]
    # phase 3: head output per lane, then the (given) dout
    dz2 = {}
    gH = {w: np.zeros((64, 4)) for w in W}
    for w in W:
        fb = 16 * w
        if actor:
            a0, a1 = gq, 4 + gq
            dout0, dout1 = [], []
            for b in range(2):
                s = 16 * b + n
                mu0 = sum(PP[((ww * 2 + b) * 8 + a0) * 16 + n] for ww in range(4))
                mu1 = sum(PP[((ww * 2 + b) * 8 + a1) * 16 + n] for ww in range(4))
                assert np.allclose(mu0[a0 < ACT], head[s, np.minimum(a0, ACT - 1)][a0 < ACT]) and np.allclose(mu0[a0 >= ACT], 0)
                assert np.allclose(mu1[a1 < ACT], head[s, np.minimum(a1, ACT - 1)][a1 < ACT]) and np.allclose(mu1[a1 >= ACT], 0)
                d0 = np.where(a0 < ACT, dout_true[s, np.minimum(a0, ACT - 1)], 0.0)
                d1 = np.where(a1 < ACT, dout_true[s, np.minimum(a1, ACT - 1)], 0.0)
                dout0.append(d0); dout1.append(d1)
                R1[s * PS + fb + a0] = d0
                R1[s * PS + fb + a1] = d1
            for J in range(2):
                bv = np.stack([R2[(fb + n) * PF + 16 * J + 4 * gq + r] for r in range(4)], 1)
                for r in range(4):
                    gH[w] = mfma16(R1[(16 * J + 4 * gq + r) * PS + fb + (n & 7)], bv[:, r], gH[w])
            dz2[w] = []
            for b in range(2):
                dh = mfma16(WHb[w][0], dout0[b], np.zeros((64, 4)))
                dh = mfma16(WHb[w][1], dout1[b], dh)
                dz2[w].append(dh * (1 - h2[w][b] ** 2))
        else:
            dz2[w] = []
            for b in range(2):
                s = 16 * b + n
                value = sum(PP[(ww * 2 + b) * 16 + n] for ww in range(4))
                assert np.allclose(value, head[s, 0])
                dout = dout_true[s, 0]
                dh = np.stack([dout * WH[w][r] for r in range(4)], 1)
                for r in range(4):
                    gH[w][:, r] += dout * h2[w][b][:, r]
                dz2[w].append(dh * (1 - h2[w][b] ** 2))
    for w in W:       # stores behind the head gradient (own columns / rows)
        fb = 16 * w
        for b in range(2):
            for r in range(4):
                R1[(16 * b + n) * PS + fb + 4 * gq + r] = dz2[w][b][:, r]
                R2[(fb + 4 * gq + r) * PF + 16 * b + n] = dz2[w][b][:, r]
    assert np.allclose(R1.reshape(32, PS)[:, :64], dZ2)
    assert np.allclose(R2.reshape(64, PF)[:, :32], dZ2.T)
    # phase 4
    NB1 = (K1 + 15) // 16
    slab = {}
    for w in W:
        fb = 16 * w
        acc = [np.zeros((64, 4)), np.zeros((64, 4))]
        for jj in range(4):
            for b in range(2):
                bv = np.stack([R1[(16 * b + n) * PS + 16 * jj + 4 * gq + r] for r in range(4)], 1)
                for r in range(4):
                    acc[b] = mfma16(W2t[w][4 * jj + r], bv[:, r], acc[b])
        for b in range(2):
            hv = np.stack([R3[(fb + 4 * gq + r) * PF + 16 * b + n] for r in range(4)], 1)
            acc[b] = acc[b] * (1 - hv ** 2)
            for r in range(4):
                assert np.allclose(acc[b][:, r], dZ1[16 * b + n, fb + 4 * gq + r])
        gW2 = [np.zeros((64, 4)) for _ in range(4)]
        rs = np.zeros(64)
        for J in range(2):
            av = np.stack([R2[(fb + n) * PF + 16 * J + 4 * gq + r] for r in range(4)], 1)
            rs += av.sum(1)
            for c in range(4):
                bv = np.stack([R3[(16 * c + n) * PF + 16 * J + 4 * gq + r] for r in range(4)], 1)
                for r in range(4):
                    gW2[c] = mfma16(av[:, r], bv[:, r], gW2[c])
        for b in range(2):
            for r in range(4):
                R2[(fb + 4 * gq + r) * PF + 16 * b + n] = acc[b][:, r]
        gW1 = [np.zeros((64, 4)) for _ in range(NB1)]
        rs1 = np.zeros(64)
        for J in range(2):
            av = np.stack([R2[(fb + n) * PF + 16 * J + 4 * gq + r] for r in range(4)], 1)
            rs1 += av.sum(1)
            for c in range(NB1):
                k = np.minimum(16 * c + n, REC_W - 1)
                bv = np.stack([REC[(16 * J + 4 * gq + r) * REC_W + k] for r in range(4)], 1)      # field k of four samples
                for r in range(4):
                    gW1[c] = mfma16(av[:, r], bv[:, r], gW1[c])
        slab[w] = (gW2, gW1, rs, rs1)
    # epilogue into a Slab3-shaped vector, then back through slab3_col_to_param's mapping
    W2T = np.zeros((64, 64)); W1T = np.zeros((K1, 64)); B2g = np.zeros(64); B1g = np.zeros(64)
    HEAD = np.zeros((64, 8)) if actor else np.zeros(64)
    for w in W:
        fb = 16 * w
        gW2, gW1, rs, rs1 = slab[w]
        for c in range(4):
            for r in range(4):
                W2T[16 * c + n, fb + 4 * gq + r] = gW2[c][:, r]
        for c in range(NB1):
            ok = 16 * c + n < K1
            for r in range(4):
                W1T[(16 * c + n)[ok], (fb + 4 * gq + r)[ok]] = gW1[c][ok, r]
        rsum = rs + rs[LANE ^ 16]; rsum = rsum + rsum[LANE ^ 32]
        B2g[(fb + n)[gq == 0]] = rsum[gq == 0]
        r1 = rs1 + rs1[LANE ^ 16]; r1 = r1 + r1[LANE ^ 32]
        B1g[(fb + n)[gq == 0]] = r1[gq == 0]
        if actor:
            sel = gq < 2
            for r in range(4):
                HEAD[(fb + n)[sel], (4 * gq + r)[sel]] = gH[w][sel, r]
        else:
            g = gH[w].copy()
            for sh in (1, 2, 4, 8):
                g = g + g[LANE ^ sh]
            sel = n == 0
            for r in range(4):
                HEAD[(fb + 4 * gq + r)[sel]] = g[sel, r]
    assert np.allclose(W2T.T, dW2)                        # slab [f1][f2] -> W2[f2][f1]
    assert np.allclose(W1T.T[:, :OBS], dW1aug[:, :OBS])   # slab [k][f1]; columns >= obs are not parameters
    assert np.allclose(B1g, dW1aug[:, OBS])
    assert np.allclose(B2g, db2)
    if actor:
        assert np.allclose(HEAD[:, :ACT].T, dWH) and np.allclose(HEAD[:, ACT:], 0)
    else:
        assert np.allclose(HEAD, dWH[0])
    print("actor" if actor else "critic", "tile: layouts consistent")


run(True)
run(False)
