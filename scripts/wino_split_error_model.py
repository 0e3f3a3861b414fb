"""Error model of the split-operand Winograd convolutions (round 5): fp32 operands split exactly into three bfloat16 terms
(x = h + m + l), the leading SIX cross products (hh, hm, mh, mm, hl, lh: everything down to 2^-16 relative; the dropped ml, lm, ll
are <= 2^-24 -- below an fp32 product's own rounding) multiplied as bf16 x bf16 -> fp32 (exact) and accumulated in fp32, against
plain fp32 and float64.  F(4x4,3x3) at the kernel's interpolation points (0, 1, -1, 2, -1/2, inf) and the F(3x3,2x2) weight-gradient
GEMM.  CPU only (NumPy);  python scripts/wino_split_error_model.py"""
import numpy as np

rng = np.random.RandomState(0)


def cook_toom(m, r, pts):
    a = m + r - 1
    P = np.array(pts, np.float64)
    AT = np.zeros((m, a)); G = np.zeros((a, r)); BT = np.zeros((a, a))
    for j in range(a - 1):
        N = np.prod([P[j] - P[l] for l in range(a - 1) if l != j])
        for i in range(m):
            AT[i, j] = P[j] ** i
        for k in range(r):
            G[j, k] = P[j] ** k / N
        poly = np.poly1d([1.0])
        for l in range(a - 1):
            if l != j:
                poly = poly * np.poly1d([1.0, -P[l]])
        c = poly.coeffs[::-1]
        BT[j, :len(c)] = c
    AT[m - 1, a - 1] = 1.0
    G[a - 1, r - 1] = 1.0
    poly = np.poly1d([1.0])
    for l in range(a - 1):
        poly = poly * np.poly1d([1.0, -P[l]])
    c = poly.coeffs[::-1]
    BT[a - 1, :len(c)] = c
    return BT, G, AT


def trunc_bf16(x):          # value of the top 16 bits (what `x & 0xffff0000` keeps)
    return (np.ascontiguousarray(x, np.float32).view(np.uint32) & np.uint32(0xffff0000)).view(np.float32)


def rne_bf16(x):
    u = np.ascontiguousarray(x, np.float32).view(np.uint32)
    r = ((u >> 16) & 1) + np.uint32(0x7fff)
    return ((u + r) & np.uint32(0xffff0000)).view(np.float32)


def split3(x, rnd):
    h = rnd(x); r1 = (x - h).astype(np.float32); m = rnd(r1); r2 = (r1 - m).astype(np.float32); l = rnd(r2)
    return h, m, l


def mm_split(A, B, terms, rnd=trunc_bf16):
    """A (M,K) @ B (K,N) with both operands split; fp32 accumulation of the chosen cross products, smallest first or last
    does not matter to first order -- the MFMA chain adds them in issue order: largest first here (hh first)."""
    a, b = split3(A, rnd), split3(B, rnd)
    acc = np.zeros((A.shape[0], B.shape[1]), np.float32)
    for i, j in terms:
        acc = acc + (a[i] @ b[j]).astype(np.float32)
    return acc


X3 = [(0, 0), (0, 1), (1, 0)]
X6 = X3 + [(1, 1), (0, 2), (2, 0)]
X9 = X6 + [(1, 2), (2, 1), (2, 2)]


def direct(x, w, dt):
    H, W, C = x.shape; K = w.shape[3]
    xp = np.zeros((H + 2, W + 2, C), dt); xp[1:-1, 1:-1] = x
    y = np.zeros((H, W, K), dt)
    for kh in range(3):
        for kw in range(3):
            y += (xp[kh:kh + H, kw:kw + W].reshape(-1, C).astype(dt) @ w[kh, kw].astype(dt)).reshape(H, W, K)
    return y


def wino(x, w, BT, G, AT, m, dt, matmul):
    H, W, C = x.shape; K = w.shape[3]; a = m + 2
    TY, TX = -(-H // m), -(-W // m)
    xp = np.zeros((TY * m + 2, TX * m + 2, C), dt); xp[1:H + 1, 1:W + 1] = x
    U = np.einsum('ik,klcn,jl->ijcn', G.astype(dt), w.astype(dt), G.astype(dt)).astype(dt)
    d = np.zeros((TY, TX, a, a, C), dt)
    for i in range(a):
        for j in range(a):
            d[:, :, i, j] = xp[i:i + TY * m:m, j:j + TX * m:m][:TY, :TX]
    V = np.einsum('xi,tuijc,nj->tuxnc', BT.astype(dt), d, BT.astype(dt)).astype(dt)
    M = np.zeros((TY, TX, a, a, K), dt)
    for xi in range(a):
        for nu in range(a):
            M[:, :, xi, nu] = matmul(V[:, :, xi, nu].reshape(-1, C), U[xi, nu]).reshape(TY, TX, K)
    Y = np.einsum('yx,tuxnk,zn->tuyzk', AT.astype(dt), M, AT.astype(dt)).astype(dt)
    return Y.transpose(0, 2, 1, 3, 4).reshape(TY * m, TX * m, K)[:H, :W]


if __name__ == '__main__':
    BT4, G4, AT4 = cook_toom(4, 3, [0, 1, -1, 2, -0.5])
    BT2, G2, AT2 = cook_toom(2, 3, [0, 1, -1])
    f32 = lambda A, B: (A.astype(np.float32) @ B.astype(np.float32))
    print('# forward / data gradient: max |err| / output range (rms err / rms) against float64 direct convolution')
    for (H, W, C, K) in [(28, 28, 256, 64), (28, 28, 512, 64), (56, 56, 64, 64), (32, 24, 512, 64)]:
        x = np.maximum(rng.randn(H, W, C), 0).astype(np.float32)
        w = (rng.randn(3, 3, C, K) * np.sqrt(2.0 / (9 * C))).astype(np.float32)
        ref = direct(x, w, np.float64)
        rv = np.abs(ref).max()
        chk = wino(x, w, BT4, G4, AT4, 4, np.float64, lambda A, B: A @ B)
        assert np.abs(chk - ref).max() < 1e-9 * rv
        rows = [('direct fp32', direct(x, w, np.float32)),
                ('F(2x2) fp32', wino(x, w, BT2, G2, AT2, 2, np.float32, f32)),
                ('F(4x4) fp32', wino(x, w, BT4, G4, AT4, 4, np.float32, f32))]
        for nm, terms in (('x3', X3), ('x6', X6), ('x9', X9)):
            rows.append(('F(4x4) bf16%s trunc' % nm, wino(x, w, BT4, G4, AT4, 4, np.float32, lambda A, B, t=terms: mm_split(A, B, t))))
        rows.append(('F(4x4) bf16x6 rne', wino(x, w, BT4, G4, AT4, 4, np.float32, lambda A, B: mm_split(A, B, X6, rne_bf16))))
        rows.append(('F(2x2) bf16x6 trunc', wino(x, w, BT2, G2, AT2, 2, np.float32, lambda A, B: mm_split(A, B, X6))))
        for nm, y in rows:
            e = np.abs(y - ref)
            print('%3dx%-3d C%-4d %-22s %.2e  (%.2e)' % (H, W, C, nm, e.max() / rv, np.sqrt((e ** 2).mean()) / np.sqrt((ref ** 2).mean())))
    print('# weight-gradient GEMM dU[c][k] = sum_t V[t][c] Z[t][k]: max |err| / range against float64')
    for T in (12544, 200704):
        V = (np.maximum(rng.randn(T, 64), 0) + np.maximum(rng.randn(T, 64), 0) - np.maximum(rng.randn(T, 64), 0)).astype(np.float32)
        Z = (rng.randn(T, 64) * 1e-4).astype(np.float32)
        ref = V.astype(np.float64).T @ Z.astype(np.float64)
        rv = np.abs(ref).max()

        def blocked(mm, blk=4096):      # the kernel's split-K: fp32 partials per block of tiles, summed in order
            acc = np.zeros((64, 64), np.float32)
            for s in range(0, T, blk):
                acc = acc + mm(np.ascontiguousarray(V[s:s + blk].T), Z[s:s + blk])
            return acc
        for nm, mm in (('fp32', f32), ('bf16x3', lambda A, B: mm_split(A, B, X3)), ('bf16x6', lambda A, B: mm_split(A, B, X6)),
                       ('bf16x9', lambda A, B: mm_split(A, B, X9))):
            e = np.abs(blocked(mm) - ref)
            print('T %-7d %-8s %.2e  (%.2e)' % (T, nm, e.max() / rv, np.sqrt((e ** 2).mean()) / np.sqrt((ref ** 2).mean())))
