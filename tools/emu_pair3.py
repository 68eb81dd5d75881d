"""CPU emulation of the data layouts of csrc/sta_xattn_proj3.hip (head-pair projection-fused forward, second generation).

Not a test of the GPU code: it checks the INDEX MATH of the design before a kernel is written against it — the packed
K / V^T block layout, the per-lane LDS addresses, the k-slot <-> head-dim / key permutations, and that chaining the
lane-level semantics of v_mfma_f32_16x16x32 / 16x16x16 through projection -> S^T -> softmax -> PV reproduces plain
attention for BOTH heads of a pair. Every formula below is mirrored one to one in the .hip file.

MFMA lane semantics (lane = 16 g + c):
  16x16x32:  A[i = c][k = 8g + j], B[k = 8g + j][col = c], j < 8;   D[i = 4g + r][col = c], r < 4
  16x16x16:  A[i = c][k = 4g + j], B[k = 4g + j][col = c], j < 4;   D as above
"""
import numpy as np

D = 40
KROW = 96            # bytes of a K row: 64 (big: 4 chunks of 16 B) + 32 (small: 4 units of 8 B)
KR = 77              # K rows stored
KBYTES = KR * KROW   # 7392
VROW = 160           # bytes of a V^T row: 2 x 64 (big steps) + 32 (small)
VR = D + 1           # 40 dims + ones row
VBYTES = VR * VROW   # 6560
BLK = KBYTES + VBYTES


def vrow_dim(r, hp):
    """Head dim held by row r of the V^T image (row 40: the ones row, -1). The rows of O^T = V^T P^T come out of the MFMAs
    as lane (g, c) <- rows 16u + 4g + {0..3}: the permutation makes a lane's tile-0 and tile-1 registers 8 CONSECUTIVE dims
    (one 16-byte store without a cross-lane exchange); head B is rotated by one lane row so that the pair's second 64-byte
    store instruction covers bytes 64..127 of the pair segment with [A tail | B dims 0..23] in lane rows 0 | 1, 2, 3."""
    if r == 40:
        return -1
    u, c = r // 16, r % 16
    g, q = c >> 2, c & 3
    if u == 2:
        return 32 + c                       # rows 32..39 (lane rows 0, 1)
    return 8 * ((g + 3 * hp) & 3) + 4 * u + q


def swz_big(g, row):      # csrc/sta_xattn_proj3.h: slot of lane row g's 16-byte chunk inside row `row`
    return g ^ ((row >> 2) & 1)


def swz_small(g, row):
    return g ^ ((row >> 2) & 3)


def pack_block(k, v, hp, M):
    """k, v: [M, 40] float16 of ONE head; hp = head parity inside its pair. Returns BLK bytes (as uint16 halves)."""
    blk = np.zeros(BLK // 2, dtype=np.float16)
    for key in range(KR):
        row = key * (KROW // 2)
        for g in range(4):
            for j in range(4):
                if hp == 0:
                    lo, hi = 4 * g + j, 16 + 4 * g + j
                    sm = 32 + 4 * g + j if g < 2 else -1
                else:
                    lo, hi = 8 + 4 * g + j, 24 + 4 * g + j
                    sm = 4 * (g - 2) + j if g >= 2 else -1
                if key < M:
                    blk[row + 8 * swz_big(g, key) + j] = k[key, lo]
                    blk[row + 8 * swz_big(g, key) + 4 + j] = k[key, hi]
                    if sm >= 0:
                        blk[row + 32 + 4 * swz_small(g, key) + j] = k[key, sm]
    vb = KBYTES // 2
    for r in range(VR):
        row = vb + r * (VROW // 2)
        dim = vrow_dim(r, hp)
        for s in range(2):
            for g in range(4):
                for j in range(8):
                    key = 32 * s + 16 * (j >> 2) + 4 * g + (j & 3)
                    if key < M:
                        blk[row + 32 * s + 8 * swz_big(g, r) + j] = v[key, dim] if dim >= 0 else 1.0
        for g in range(4):
            for j in range(4):
                key = 64 + 4 * g + j
                if key < M:
                    blk[row + 64 + 4 * swz_small(g, r) + j] = v[key, dim] if dim >= 0 else 1.0
    return blk


def permlane16_swap(x, y):
    """v_permlane16_swap vdst=x, src=y on [64 lanes] arrays: odd rows of x <-> even rows of y. Returns (x', y')."""
    x2, y2 = x.copy(), y.copy()
    for row in (1, 3):
        x2[16 * row:16 * row + 16] = y[16 * (row - 1):16 * (row - 1) + 16]
        y2[16 * (row - 1):16 * (row - 1) + 16] = x[16 * row:16 * row + 16]
    return x2, y2


def store_pair(oA, oB):
    """oA, oB: per head [3 tiles][64 lanes][4] normalised O^T accumulators of ONE batch row. Returns the pair segment
    [16 px][80 dims] as the three store instructions of the kernel write it (16-byte pieces = 8 dims per lane)."""
    seg = np.full((16, 80), np.nan)
    def put(lane_vals, byte_off_of_lane, active):
        for lane in range(64):
            if active(lane):
                c = lane & 15
                b = byte_off_of_lane(lane)
                seg[c, b // 2:b // 2 + 8] = lane_vals[lane]
    mainA = np.concatenate([oA[0], oA[1]], axis=1)        # [64][8]: lane's tile-0 | tile-1 registers
    mainB = np.concatenate([oB[0], oB[1]], axis=1)
    # tails: lane rows 0, 1 hold dims 32..35 / 36..39 of their head in tile 2; one swap per dword pair hands lane row 0 the A
    # tail (own | row 1's) and lane row 1 the B tail
    tail = np.zeros((64, 8))
    for d in range(4):
        x, y = permlane16_swap(oA[2][:, d], oB[2][:, d])
        tail[:, d], tail[:, 4 + d] = x, y
    g_of = lambda lane: lane >> 4
    put(mainA, lambda l: 16 * g_of(l), lambda l: True)                                    # I1: bytes 0..63
    v2 = np.where((np.arange(64) >> 4 == 0)[:, None], tail, mainB)
    put(v2, lambda l: 64 + 16 * g_of(l), lambda l: True)                                  # I2: bytes 64..127
    v3 = np.where((np.arange(64) >> 4 == 0)[:, None], mainB, tail)
    put(v3, lambda l: 128 + 16 * g_of(l), lambda l: g_of(l) < 2)                          # I3: bytes 128..159
    return seg


def mfma32(A, B, Cacc):
    """A: [64 lanes][8], B: [64][8], C: [64][4] -> D[64][4] with the lane semantics above."""
    a = np.zeros((16, 32)); b = np.zeros((32, 16))
    for lane in range(64):
        g, c = lane >> 4, lane & 15
        a[c, 8 * g:8 * g + 8] = A[lane]
        b[8 * g:8 * g + 8, c] = B[lane]
    d = a @ b
    out = Cacc.copy()
    for lane in range(64):
        g, c = lane >> 4, lane & 15
        out[lane] += d[4 * g:4 * g + 4, c]
    return out


def mfma16(A, B, Cacc):
    a = np.zeros((16, 16)); b = np.zeros((16, 16))
    for lane in range(64):
        g, c = lane >> 4, lane & 15
        a[c, 4 * g:4 * g + 4] = A[lane]
        b[4 * g:4 * g + 4, c] = B[lane]
    d = a @ b
    out = Cacc.copy()
    for lane in range(64):
        g, c = lane >> 4, lane & 15
        out[lane] += d[4 * g:4 * g + 4, c]
    return out


def lds_read(lds, byte_off, nbytes):
    assert byte_off % 2 == 0
    return lds[byte_off // 2: byte_off // 2 + nbytes // 2].astype(np.float64)


def emulate(seed=0, M=77, C=320):
    rng = np.random.default_rng(seed)
    H = C // D
    pr = 1                                       # pair index -> heads 2, 3
    y = rng.standard_normal((16, C)).astype(np.float16)             # one wave's 16 pixels, one batch row
    wq = (rng.standard_normal((C, C)) / np.sqrt(C)).astype(np.float16)
    k = (rng.standard_normal((M, C)) * 0.7).astype(np.float16)
    v = rng.standard_normal((M, C)).astype(np.float16)
    scale = D ** -0.5
    # ---- projection: pair columns 80 pr .. +79, 5 tiles; acc[u][lane][r] = Q^T[col 16u + 4g + r][px c] ----------
    nkc = C // 32
    acc = [np.zeros((64, 4)) for _ in range(5)]
    for s in range(nkc):
        B = np.zeros((64, 8))
        for lane in range(64):
            g, c = lane >> 4, lane & 15
            B[lane] = y[c, 32 * s + 8 * g: 32 * s + 8 * g + 8]
        for u in range(5):
            A = np.zeros((64, 8))
            for lane in range(64):
                g, c = lane >> 4, lane & 15
                A[lane] = wq[80 * pr + 16 * u + c, 32 * s + 8 * g: 32 * s + 8 * g + 8]
            acc[u] = mfma32(A, B, acc[u])
    q16 = [a.astype(np.float16).astype(np.float64) for a in acc]      # rounded to T once
    qbig = {0: np.concatenate([q16[0], q16[1]], axis=1), 1: np.concatenate([q16[3], q16[4]], axis=1)}
    qsm = q16[2]
    qref = (y.astype(np.float64) @ wq.astype(np.float64).T).astype(np.float16).astype(np.float64)
    worst = 0.0
    o_norm, refs = [], []
    for hp in range(2):
        h = 2 * pr + hp
        lds = pack_block(k[:, h * D:(h + 1) * D], v[:, h * D:(h + 1) * D], hp, M)
        # over-read area behind the block: finite garbage
        lds = np.concatenate([lds, rng.standard_normal(1024).astype(np.float16)])
        # ---- S^T = K Q^T ----------------------------------------------------------------------------------------
        st = []
        for t in range(5):
            Abig = np.zeros((64, 8)); Asm = np.zeros((64, 4))
            for lane in range(64):
                g, c = lane >> 4, lane & 15
                Abig[lane] = lds_read(lds, c * KROW + 16 * swz_big(g, c) + t * 16 * KROW, 16)
                Asm[lane] = lds_read(lds, c * KROW + 64 + 8 * swz_small(g, c) + t * 16 * KROW, 8)
            a = np.zeros((64, 4))
            if t == 4:
                for lane in range(64):
                    g = lane >> 4
                    for r in range(4):
                        if 64 + 4 * g + r >= M:
                            a[lane, r] = -1.0e30
            a = mfma32(Abig, qbig[hp], a)
            a = mfma16(Asm, qsm, a)
            st.append(a)
        # ---- softmax over keys (all 4 lane rows of a pixel column) -------------------------------------------
        S = np.zeros((80, 16))
        for t in range(5):
            for lane in range(64):
                g, c = lane >> 4, lane & 15
                S[16 * t + 4 * g:16 * t + 4 * g + 4, c] = st[t][lane]
        mx = S.max(axis=0, keepdims=True)
        P = np.exp2((S - mx) * scale * 1.4426950408889634)
        p_t = []
        for t in range(5):
            pt = np.zeros((64, 4))
            for lane in range(64):
                g, c = lane >> 4, lane & 15
                pt[lane] = P[16 * t + 4 * g:16 * t + 4 * g + 4, c]
            p_t.append(pt.astype(np.float16).astype(np.float64))
        pbig = [np.concatenate([p_t[0], p_t[1]], axis=1), np.concatenate([p_t[2], p_t[3]], axis=1)]
        psm = p_t[4]
        # ---- O^T = V^T P^T ---------------------------------------------------------------------------------------
        o = []
        for u in range(3):
            a = np.zeros((64, 4))
            for s in range(2):
                A = np.zeros((64, 8))
                for lane in range(64):
                    g, c = lane >> 4, lane & 15
                    A[lane] = lds_read(lds, KBYTES + (16 * u + c) * VROW + 64 * s + 16 * swz_big(g, c), 16)
                a = mfma32(A, pbig[s], a)
            A = np.zeros((64, 4))
            for lane in range(64):
                g, c = lane >> 4, lane & 15
                A[lane] = lds_read(lds, KBYTES + (16 * u + c) * VROW + 128 + 8 * swz_small(g, c), 8)
            a = mfma16(A, psm, a)
            o.append(a)
        den = o[2][32:48, 0]                           # ones row 40 = tile 2, row 8: lane row g = 2, register 0
        inv = np.zeros(64)
        for lane in range(64):
            inv[lane] = 1.0 / den[lane & 15]           # broadcast of lane row 2 to all lane rows
        o_norm.append([o[u] * inv[:, None] for u in range(3)])
        # ---- reference ---------------------------------------------------------------------------------------------
        qh = qref[:, h * D:(h + 1) * D]
        kh = k[:, h * D:(h + 1) * D].astype(np.float64)
        vh = v[:, h * D:(h + 1) * D].astype(np.float64)
        sim = qh @ kh.T * scale
        pr_ = np.exp(sim - sim.max(axis=1, keepdims=True))
        refs.append((pr_ / pr_.sum(axis=1, keepdims=True)) @ vh)
    seg = store_pair(o_norm[0], o_norm[1])
    assert not np.isnan(seg).any(), "a byte of the pair segment was not written"
    ref = np.concatenate(refs, axis=1)                 # [16 px][80 dims]: head A | head B
    return np.abs(seg - ref).max()


if __name__ == "__main__":
    for M in (77, 70, 65):
        e = emulate(seed=M, M=M)
        print("M=%d max |out - ref| = %.3e" % (M, e))
        assert e < 3e-3, e
    print("emulation ok: BLK=%d bytes, LDS for K=2: %d" % (BLK, 8 * BLK + 5 * 10 * 1024))
