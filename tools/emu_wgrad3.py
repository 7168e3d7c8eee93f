"""CPU emulation of the index arithmetic of csrc/wgrad3.hip (LDS images, swizzles, transposed-read address tables, wave ->
fragment ownership, slab layout) for one launch, checked against a direct numpy weight gradient.  Development tool: it mirrors
the kernel line by line so that an indexing change can be validated without a GPU round trip.

    python tools/emu_wgrad3.py
"""
import itertools
import numpy as np


def geom(B, H, W, nchunks, N):
    best = None
    for tw in (8, 16, 32):
        thmax = 128 // tw
        tilesX = -(-W // tw)
        tilesY = -(-H // thmax)
        th = -(-H // tilesY)
        hwc = min(tw, W) + 2
        if (thmax + 2) * hwc > 256:
            continue
        key = tilesX * tilesY * 100000 + (thmax + 2) * hwc
        if best is None or key <= best[0]:
            best = (key, dict(TW=tw, TH=th, HWc=hwc, tilesX=tilesX, tilesY=tilesY))
    g = best[1]
    npad = (N + 15) // 16 * 16
    g["WC"] = 2 if nchunks == 1 else 4
    g["NTL"] = 64 if npad >= 48 else 32
    g["KT"] = -(-nchunks // (g["WC"] // 2))
    g["NTt"] = -(-npad // g["NTL"])
    g["patches"] = B * g["tilesX"] * g["tilesY"]
    return g


def run(B, H, W, cs, N, nsplit_want=3, affine=False, seed=0):
    rng = np.random.default_rng(seed)
    K = sum(cs)
    nchunks = K // 32
    xs = [rng.standard_normal((B, H, W, c)).astype(np.float32) for c in cs]
    dy = rng.standard_normal((B, H, W, N)).astype(np.float32)
    sc = (1 + 0.3 * rng.standard_normal(K)).astype(np.float32)
    sh = (0.2 * rng.standard_normal(K)).astype(np.float32)
    chunk = []   # (src, c0)
    for si, c in enumerate(cs):
        for c0 in range(0, c, 32):
            chunk.append((si, c0))
    g = geom(B, H, W, nchunks, N)
    TW, TH, HWc = g["TW"], g["TH"], g["HWc"]
    WC, NTL = g["WC"], g["NTL"]
    CPT = WC // 2
    NFT = NTL // 16
    WN = 4 // WC
    NF = NFT // WN
    NPL = (NFT + 1) // 2
    P = TH * TW
    HPv = (TH + 2) * HWc
    xpl = (128 // TW + 2) * HWc * 64
    stage = CPT * xpl + NPL * 8192
    lds_bytes = 2 * stage + 1024
    kstride = (32 // TW) * HWc * 64
    hymask = 1 if TW == 8 else 0
    npad = (N + 15) // 16 * 16
    Ktot = nchunks * 32
    patches = g["patches"]
    want = min(nsplit_want, patches)
    pps = -(-patches // want)
    nsplit = -(-patches // pps)
    partial = np.full((nsplit, 9, Ktot, npad), np.nan, np.float32)

    def x_of(ch, b, iy, ix, c):      # channel c (0..31) of chunk ch
        si, c0 = chunk[ch]
        v = xs[si][b, iy, ix, c0 + c]
        if affine:
            kk = sum(cs[:si]) + c0 + c
            v = max(v * sc[kk] + sh[kk], 0.0)
        return v

    for split, kt, nt in itertools.product(range(nsplit), range(g["KT"]), range(g["NTt"])):
        n0 = nt * NTL
        lds = np.zeros(lds_bytes // 2, np.float32)       # one entry per bf16 element
        cvalid = [kt * CPT + c < nchunks for c in range(CPT)]
        acc = np.zeros((4, 9, NF, 64, 4), np.float32)     # wave, tap, nf, lane, reg
        p_begin, p_end = split * pps, min(patches, split * pps + pps)

        def issue_loads(patch, stg):
            q1 = patch // g["tilesX"]; tx = patch - q1 * g["tilesX"]
            b = q1 // g["tilesY"]; ty = q1 - b * g["tilesY"]
            oy0, ox0 = ty * TH, tx * TW
            sb = stg * stage
            for tid in range(256):
                for it in range(4):
                    v = it * 256 + tid
                    R, s = v >> 2, v & 3
                    hy, hx = R // HWc, R % HWc
                    iy, ix = oy0 - 1 + hy, ox0 - 1 + hx
                    inr = v < HPv * 4
                    ok = inr and 0 <= iy < H and 0 <= ix < W
                    f = ((hx >> 3) ^ (hy & hymask)) & 1
                    sl = s ^ (f << 1)
                    if it * 256 < HPv * 4:
                        for c in range(CPT):
                            if not cvalid[c]:
                                continue
                            dst = (sb + c * xpl + v * 16) // 2       # lane-linear destination
                            if ok:
                                for e in range(8):
                                    lds[dst + e] = x_of(kt * CPT + c, b, iy, ix, sl * 8 + e)
                            elif inr:
                                lds[dst:dst + 8] = 0
                for it in range(2):
                    v = it * 256 + tid
                    p, s = v >> 2, v & 3
                    ly, lx = p // TW, p % TW
                    pv = p < P and oy0 + ly < H and ox0 + lx < W
                    sl = s ^ (((p >> 3) & 1) << 1)
                    for j in range(NPL):
                        n = n0 + j * 32 + sl * 8
                        dst = (sb + CPT * xpl + j * 8192 + v * 16) // 2
                        if pv and n < N:
                            lds[dst:dst + 8] = dy[b, oy0 + ly, ox0 + lx, n:n + 8]
                        else:
                            lds[dst:dst + 8] = 0

        def tr_frag(addr):            # addr[lane] byte addresses (h = 0 / 1 handled by the caller); returns [lane][4]
            out = np.zeros((64, 4), np.float32)
            for lane in range(64):
                gb, i = lane & ~15, lane & 15
                for j in range(4):
                    src_lane = gb + 4 * j + (i >> 2)
                    out[lane, j] = lds[addr[src_lane] // 2 + (i & 3)]
            return out

        # tables
        a_tab = np.zeros((4, 2, 3, 2, 64), np.int64)
        b_tab = np.zeros((4, NF, 2, 64), np.int64)
        for wave in range(4):
            cfi, wn = wave % WC, wave // WC
            for lane in range(64):
                gq, l15 = lane >> 4, lane & 15
                for h in range(2):
                    p0 = gq * 8 + h * 4 + (l15 >> 2)
                    ly, lx = p0 // TW, p0 % TW
                    for ty in range(2):
                        for tx in range(3):
                            hy, hx = ly + ty, lx + tx
                            f = ((hx >> 3) ^ (hy & hymask)) & 1
                            a_tab[wave, ty, tx, h, lane] = (cfi >> 1) * xpl + (hy * HWc + hx) * 64 + (((cfi & 1) ^ f) << 5) + (l15 & 3) * 8
                    for nf in range(NF):
                        nfg = wn * NF + nf
                        fd = (p0 >> 3) & 1
                        b_tab[wave, nf, h, lane] = CPT * xpl + (nfg >> 1) * 8192 + p0 * 64 + (((nfg & 1) ^ fd) << 5) + (l15 & 3) * 8
        row2 = 2 * HWc * 64

        if p_begin < p_end:
            issue_loads(p_begin, 0)
        for patch in range(p_begin, p_end):
            stg = (patch - p_begin) & 1
            if patch + 1 < p_end:
                issue_loads(patch + 1, stg ^ 1)
            so = stg * stage
            for wave in range(4):
                cfi = wave % WC
                if not cvalid[cfi >> 1]:
                    continue
                for ks in range(4):
                    ao, bo = so + ks * kstride, so + ks * 2048
                    bf = []
                    for nf in range(NF):
                        lo, hi = tr_frag(b_tab[wave, nf, 0] + bo), tr_frag(b_tab[wave, nf, 1] + bo)
                        bf.append(np.concatenate([lo, hi], 1))          # [lane][8]
                    for t in range(9):
                        ty, tx = t // 3, t % 3
                        o = ao + (row2 if ty == 2 else 0)
                        lo, hi = tr_frag(a_tab[wave, ty & 1, tx, 0] + o), tr_frag(a_tab[wave, ty & 1, tx, 1] + o)
                        af = np.concatenate([lo, hi], 1)
                        # MFMA 16x16x32: D[i][j] += sum_k A[i][k] B[j][k]; A row i = lane&15, k = (lane>>4)*8 + e
                        A = np.zeros((16, 32), np.float32)
                        for lane in range(64):
                            A[lane & 15, (lane >> 4) * 8:(lane >> 4) * 8 + 8] = af[lane]
                        for nf in range(NF):
                            Bm = np.zeros((16, 32), np.float32)
                            for lane in range(64):
                                Bm[lane & 15, (lane >> 4) * 8:(lane >> 4) * 8 + 8] = bf[nf][lane]
                            D = A @ Bm.T
                            for lane in range(64):
                                for r in range(4):
                                    acc[wave, t, nf, lane, r] += D[(lane >> 4) * 4 + r, lane & 15]
        for wave in range(4):
            cfi, wn = wave % WC, wave // WC
            if not cvalid[cfi >> 1]:
                continue
            for lane in range(64):
                gq, l15 = lane >> 4, lane & 15
                krow0 = (kt * CPT + (cfi >> 1)) * 32 + (cfi & 1) * 16 + gq * 4
                for t in range(9):
                    for nf in range(NF):
                        n = n0 + (wn * NF + nf) * 16 + l15
                        if n >= npad:
                            continue
                        for r in range(4):
                            partial[split, t, krow0 + r, n] = acc[wave, t, nf, lane, r]
    assert not np.isnan(partial).any(), "slab elements never written"
    got = partial.sum(0)[:, :, :N]                      # [tap][k][n]
    # reference
    xc = np.concatenate(xs, -1)
    if affine:
        xc = np.maximum(xc * sc + sh, 0)
    xp = np.pad(xc, ((0, 0), (1, 1), (1, 1), (0, 0)))
    ref = np.zeros((9, K, N), np.float32)
    for t in range(9):
        ty, tx = t // 3, t % 3
        ref[t] = np.einsum("bhwk,bhwn->kn", xp[:, ty:ty + H, tx:tx + W, :], dy)
    err = np.abs(got - ref).max() / np.abs(ref).max()
    print(f"B={B} {H}x{W} cs={cs} N={N} TW={TW} TH={TH} HWc={HWc} WC={WC} NTL={NTL} splits={nsplit} aff={affine}: rel err {err:.2e}")
    assert err < 1e-5


if __name__ == "__main__":
    run(1, 16, 16, [64], 64)
    run(1, 8, 8, [32], 32, affine=True)
    run(1, 14, 14, [32, 64], 24)
    run(2, 16, 32, [32], 64)
    run(1, 6, 40, [64, 32, 64], 48)
    run(1, 28, 28, [64], 32, nsplit_want=4)
    run(1, 24, 8, [96], 32)
    print("ok")
