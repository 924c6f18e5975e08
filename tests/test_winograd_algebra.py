"""The algebra the Winograd kernels rely on (csrc/conv_wino_kernel.h, csrc/conv_wino_wgrad.hip), restated in numpy and checked
against the direct 3x3 correlation of oracle/np_ops.py: transform matrices, the per-wave folding of the four nu products, the
row of A^T that carries the bias, the fragment order of the transformed filter, and the sign convention of the weight
gradient's output-gradient transform.  CPU only; the kernels themselves are compared with the oracle in tests/test_gpu_ops.py."""
import numpy as np

from oracle import np_ops as N

BT = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], np.float64)
G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], np.float64)
AT = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], np.float64)


def _direct(x, w, b=None):
    return N.conv2d(x, w, b)


def test_forward_tile_identity_and_folding():
    rng = np.random.default_rng(0)
    H, W, C, K = 8, 10, 5, 3
    x = rng.standard_normal((1, H, W, C))
    w = rng.standard_normal((3, 3, C, K))
    b = rng.standard_normal(K)
    ref = _direct(x, w, b)[0]
    xp = np.pad(x[0], ((1, 1), (1, 1), (0, 0)))
    U = np.einsum('xa,abck,nb->xnck', G, w, G)                      # U[xi][nu][cin][cout]
    out = np.zeros((H, W, K))
    for ty in range(H // 2):
        for tx in range(W // 2):
            d = xp[2 * ty:2 * ty + 4, 2 * tx:2 * tx + 4]            # halo origin (y0 - 1, x0 - 1)
            V = np.einsum('xr,rcn,mc->xmn', BT, d, BT)              # V[xi][nu][cin]
            M = np.einsum('xnc,xnck->xnk', V, U)
            # what wave xi leaves in LDS: R0 = M0 + M1 + M2, R1 = M1 - M2 - M3 (nu = 3 enters through the NEGATED filter);
            # the bias rides on wave 1 (row 1 of A^T has +1 in both output rows)
            R = np.stack([M[:, 0] + M[:, 1] + M[:, 2], M[:, 1] - M[:, 2] + (V[:, 3][:, :, None] * -U[:, 3]).sum(1)], 1)
            R[1] += b
            # phase C: Y[i][j] = R[i][j] + sgn_i (R[i+1][j] + R[i+2][j])
            for i in range(2):
                out[2 * ty + i, 2 * tx:2 * tx + 2] = R[i] + (1 - 2 * i) * (R[i + 1] + R[i + 2])
            np.testing.assert_allclose(np.einsum('ix,xnk,jn->ijk', AT, M, AT) + b, out[2 * ty:2 * ty + 2, 2 * tx:2 * tx + 2], atol=1e-12)
    np.testing.assert_allclose(out, ref, atol=1e-11)


def test_filter_fragment_order_covers_every_element_once():
    # wino_filter_kernel: element ((pc * 4 + xi) * F/4 + f4) * 64 + lane, component j; f = 4 f4 + j = (nu * 4 KQ + ks) * NT + cb;
    # lane (l15, lq) -> cin = 16 (ks >> 2) + 4 lq + (ks & 3), cout = 16 cb + l15
    for KQ, NT in ((2, 2), (2, 3), (3, 2), (3, 3)):
        F = 16 * KQ * NT
        seen = set()
        for xi in range(4):
            for f in range(F):
                nu, ks, cb = f // (4 * KQ * NT), (f // NT) % (4 * KQ), f % NT
                for lane in range(64):
                    cin = 16 * (ks >> 2) + 4 * (lane >> 4) + (ks & 3)
                    cout = 16 * cb + (lane & 15)
                    seen.add((xi, nu, cin, cout))
        assert len(seen) == 16 * (16 * KQ) * (16 * NT) == 4 * F * 64


def test_weight_gradient_identity_with_positive_rows():
    rng = np.random.default_rng(1)
    H, W, C, K = 6, 8, 3, 4
    x = rng.standard_normal((1, H, W, C))
    dy = rng.standard_normal((1, H, W, K))
    # direct: dW[a][b][c][k] = sum_{y,x} xpad[y + a][x + b][c] dy[y][x][k]
    xp = np.pad(x[0], ((1, 1), (1, 1), (0, 0)))
    ref = np.zeros((3, 3, C, K))
    for a in range(3):
        for b in range(3):
            ref[a, b] = np.einsum('yxc,yxk->ck', xp[a:a + H, b:b + W], dy[0])
    A = AT.T                                                        # 4 x 2
    Apos = np.abs(A)                                                # what phase A' stores: rows (y0, y0 + y1, y0 - y1, y1)
    Apos[2] = A[2]                                                  # (row 2 keeps its minus: y0 - y1)
    sig = np.array([1, 1, 1, -1.0])
    dU = np.zeros((4, 4, C, K))
    db = np.zeros(K)
    for ty in range(H // 2):
        for tx in range(W // 2):
            d = xp[2 * ty:2 * ty + 4, 2 * tx:2 * tx + 4]
            V = np.einsum('xr,rcn,mc->xmn', BT, d, BT)
            t = dy[0, 2 * ty:2 * ty + 2, 2 * tx:2 * tx + 2]
            dMp = np.einsum('xi,ijk,nj->xnk', Apos, t, Apos)        # stored without the signs of row / column 3
            np.testing.assert_allclose(dMp * sig[:, None, None] * sig[None, :, None], np.einsum('xi,ijk,nj->xnk', A, t, A), atol=1e-12)
            dU += np.einsum('xnc,xnk->xnck', V, dMp)
            db += dMp[1, 1]                                         # = the tile's four pixels
    dU *= sig[:, None, None, None] * sig[None, :, None, None]       # wino_wgrad_finish_kernel
    dW = np.einsum('xa,xnck,nb->abck', G, dU, G)
    np.testing.assert_allclose(dW, ref, atol=1e-11)
    np.testing.assert_allclose(db, dy[0].sum((0, 1)), atol=1e-12)
