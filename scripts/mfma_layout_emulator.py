"""Lane-level NumPy emulation of v_mfma_f32_32x32x2_f32 used to validate the operand layouts,
the K-permutation chaining trick and the LDS transposes of the fused PPO kernel BEFORE writing HIP.
(scratch/design tool; not part of the product or the tests)"""
import numpy as np

rng = np.random.default_rng(0)
L = np.arange(64)
HALF = L >> 5
COL = L & 31


def F(r, h):  # feature (row) index inside a 32-row M tile held by acc reg r of a lane in half h
    return (r & 3) + 8 * (r >> 2) + 4 * h


def mfma(a, b, c):
    """a,b: [64] per-lane scalars; c: [64,16] accumulators. A[i=l&31][k=l>>5], B[k=l>>5][j=l&31],
    C[row=F(r,h)][col=l&31]."""
    A = np.zeros((32, 2)); B = np.zeros((2, 32))
    A[COL, HALF] = a
    B[HALF, COL] = b
    D = A @ B
    out = c.copy()
    for r in range(16):
        out[:, r] += D[F(r, HALF), COL]
    return out


S, OBS, H, ACT = 32, 17, 64, 6
X = rng.normal(size=(S, OBS)); W1 = rng.normal(size=(H, OBS)); b1 = rng.normal(size=H)
W2 = rng.normal(size=(H, H)); b2 = rng.normal(size=H)

# ---------------- layer 1: k(s,h) = 9h + s over Xaug (col 17 = 1)
Xaug = np.concatenate([X, np.ones((S, 1))], 1)          # [S,18]
W1aug = np.concatenate([W1, b1[:, None]], 1)            # [H,18]
xb = np.zeros((64, 9))
for s in range(9):
    xb[:, s] = Xaug[COL, 9 * HALF + s]                   # lane (j,h) holds Xaug[j][9h+s]
acc1 = [np.zeros((64, 16)) for _ in range(2)]
for t in range(2):
    for s in range(9):
        a = W1aug[32 * t + COL, 9 * HALF + s]            # lane (i,h): W1aug[32t+i][9h+s]
        acc1[t] = mfma(a, xb[:, s], acc1[t])
Z1 = Xaug @ W1aug.T                                     # [S,H]
for t in range(2):
    for r in range(16):
        assert np.allclose(acc1[t][:, r], Z1[COL, 32 * t + F(r, HALF)])
h1 = [np.tanh(a) for a in acc1]
H1 = np.tanh(Z1)

# ---------------- layer 2 chained: kstep (t,r) <-> feature 32t+F(r,h); B operand = h1[t][:,r]
acc2 = [np.zeros((64, 16)) for _ in range(2)]
for t2 in range(2):
    for r in range(16):
        acc2[t2][:, r] = b2[32 * t2 + F(r, HALF)]       # bias preloaded into the accumulator
    for t in range(2):
        for r in range(16):
            a = W2[32 * t2 + COL, 32 * t + F(r, HALF)]   # lane (i,h): W2[32t2+i][32t+F(r,h)]
            acc2[t2] = mfma(a, h1[t][:, r], acc2[t2])
Z2 = H1 @ W2.T + b2
for t in range(2):
    for r in range(16):
        assert np.allclose(acc2[t][:, r], Z2[COL, 32 * t + F(r, HALF)])
H2 = np.tanh(Z2)
h2 = [np.tanh(a) for a in acc2]

# ---------------- heads on VALU: partial over the lane's 32 features, combine halves via xor 32
Wmu = rng.normal(size=(ACT, H)); bmu = rng.normal(size=ACT)
mu_part = np.zeros((64, ACT))
for t in range(2):
    for r in range(16):
        mu_part += h2[t][:, r][:, None] * Wmu[:, 32 * t + F(r, HALF)].T
mu = mu_part + mu_part[L ^ 32] + bmu
assert np.allclose(mu, (H2 @ Wmu.T + bmu)[COL])

# ---------------- backward: dZ2 (same layout), dH1 = dZ2 @ W2 chained, dW2 via LDS transposes
dmu = rng.normal(size=(S, ACT))
dH2 = dmu @ Wmu                                            # [S,H]
dZ2 = dH2 * (1 - H2 ** 2)
dz2 = [np.zeros((64, 16)) for _ in range(2)]
for t in range(2):
    for r in range(16):
        f = 32 * t + F(r, HALF)
        dz2[t][:, r] = (dmu[COL] * Wmu[:, f].T).sum(1) * (1 - h2[t][:, r] ** 2)
        assert np.allclose(dz2[t][:, r], dZ2[COL, f])
dh1 = [np.zeros((64, 16)) for _ in range(2)]
for t1 in range(2):
    for t in range(2):
        for r in range(16):
            a = W2[32 * t + F(r, HALF), 32 * t1 + COL]    # lane (i,h): W2[32t+F(r,h)][32t1+i]
            dh1[t1] = mfma(a, dz2[t][:, r], dh1[t1])
dH1 = dZ2 @ W2
for t in range(2):
    for r in range(16):
        assert np.allclose(dh1[t][:, r], dH1[COL, 32 * t + F(r, HALF)])

# transposed LDS tiles T[f][s] (pitch irrelevant here); lane (j,h) reg (t,r) writes T[32t+F(r,h)][j]
def to_lds(regs):
    T = np.zeros((64, 32))
    for t in range(2):
        for r in range(16):
            T[32 * t + F(r, HALF), COL] = regs[t][:, r]
    return T
T_dz2, T_h1 = to_lds(dz2), to_lds(h1)
assert np.allclose(T_dz2, dZ2.T) and np.allclose(T_h1, H1.T)
# dW2[f2][f1] = sum_s dZ2[s][f2] H1[s][f1]; kstep s (0..15): half h <-> sample 16h+s
dW2 = dZ2.T @ H1
for tM in range(2):
    for tN in range(2):
        acc = np.zeros((64, 16))
        for s in range(16):
            a = T_dz2[32 * tM + COL, 16 * HALF + s]
            b = T_h1[32 * tN + COL, 16 * HALF + s]
            acc = mfma(a, b, acc)
        for r in range(16):   # acc reg r lane (jn,h) -> dW2[32tM+F(r,h)][32tN+jn]
            assert np.allclose(acc[:, r], dW2[32 * tM + F(r, HALF), 32 * tN + COL])
# dW1aug[f1][k] = sum_s dZ1[s][f1] Xaug[s][k], XT[k][s] from lane (j,h): XT[9h+s'][j] = xb[:,s']
dZ1 = dH1 * (1 - H1 ** 2)
dz1 = [dh1[t] * (1 - h1[t] ** 2) for t in range(2)]
T_dz1 = to_lds(dz1)
XT = np.zeros((32, 32))
for s in range(9):
    XT[9 * HALF + s, COL] = xb[:, s]
dW1aug = dZ1.T @ Xaug
for tM in range(2):
    acc = np.zeros((64, 16))
    for s in range(16):
        a = T_dz1[32 * tM + COL, 16 * HALF + s]
        b = XT[COL, 16 * HALF + s]
        acc = mfma(a, b, acc)
    for r in range(16):
        got = acc[:, r]
        ref = np.zeros(64)
        k = COL
        ref[k < 18] = dW1aug[32 * tM + F(r, HALF), :][np.arange(64)[k < 18] % 64 * 0 + 0] if False else 0
        for l in range(64):
            if COL[l] < 18:
                assert np.isclose(got[l], dW1aug[32 * tM + F(r, HALF[l]), COL[l]])
# dWmu[a][f] = sum_s dmu[s][a] H2[s][f]: lane = feature f, loop over samples with broadcast dmu
T_h2 = to_lds(h2)
assert np.allclose(dmu.T @ H2, np.stack([(dmu[:, a][None, :] * T_h2).sum(1) for a in range(ACT)]))
print("all layout identities hold")
