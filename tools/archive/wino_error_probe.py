"""fp32 error of Winograd F(2x2,3x3) and F(4x4,3x3) against an fp64 direct convolution (numpy, CPU): the accuracy side of
the F(4x4,3x3) lever of DESIGN.md section 11.  Data like a 64 -> 64 layer: post-ReLU inputs, He-scaled weights."""
import numpy as np
rs = np.random.RandomState(0)
C, K, H, W = 64, 64, 24, 24
x = np.maximum(rs.randn(C, H + 2, W + 2), 0).astype(np.float32)          # padded post-ReLU input
w = (rs.randn(K, C, 3, 3) * np.sqrt(2.0 / (C * 9))).astype(np.float32)
def direct(x, w, dt):
    x, w = x.astype(dt), w.astype(dt)
    out = np.zeros((K, H, W), dt)
    for ky in range(3):
        for kx in range(3):
            out += np.einsum("kc,chw->khw", w[:, :, ky, kx], x[:, ky:ky + H, kx:kx + W])
    return out
ref = direct(x, w, np.float64)
def wino(x, w, m, BT, G, AT):
    a = m + 2  # patch size
    BT, G, AT = [np.asarray(t, np.float32) for t in (BT, G, AT)]
    U = np.einsum("ij,kcjl,ml->kcim", G, w, G).astype(np.float32)            # G g G^T   [K,C,a,a]
    out = np.zeros((K, H, W), np.float32)
    for ty in range(0, H, m):
        for tx in range(0, W, m):
            d = x[:, ty:ty + a, tx:tx + a]
            V = np.einsum("ij,cjl,ml->cim", BT, d, BT).astype(np.float32)   # B^T d B
            M = np.einsum("kcij,cij->kij", U, V).astype(np.float32)         # fp32 accumulate over channels
            out[:, ty:ty + m, tx:tx + m] = np.einsum("ij,kjl,ml->kim", AT, M, AT)
    return out
BT2 = [[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]]
G2 = [[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]]
AT2 = [[1, 1, 1, 0], [0, 1, -1, -1]]
BT4 = [[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]]
G4 = [[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]]
AT4 = [[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]]
def err(o): return float(np.abs(o - ref).max() / np.abs(ref).max()), float(np.linalg.norm(o - ref) / np.linalg.norm(ref))
for tag, o in (("direct fp32", direct(x, w, np.float32)), ("F(2x2,3x3) fp32", wino(x, w, 2, BT2, G2, AT2)),
               ("F(4x4,3x3) fp32", wino(x, w, 4, BT4, G4, AT4))):
    e = err(o)
    print("%-18s max |err| / max |ref| %.2e   rel L2 %.2e" % (tag, e[0], e[1]))
