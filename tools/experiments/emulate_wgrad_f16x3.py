"""CPU emulation of the INDEX ARITHMETIC of wgrad_f16x3_kernel (csrc/wgrad.hip): LDS images (one float per fp16 slot), staging item
decoding, K half-blocks, fragment reads (ds_read_b128 + ds_read_b32), the dw funnel shift and the (lane half, element) pairing of the
MFMA -- against a direct evaluation of dW[a][b][t] = sum_pos P[pos][a] Q[pos + off_t][b].  Run before the kernel ever saw a GPU."""
import itertools
import numpy as np

def emulate(B, Pd, Ph, Pw, A, Bc, kd, kh, kw, pad, seed=0):
    rng = np.random.default_rng(seed)
    P = rng.normal(size=(B, Pd, Ph, Pw, A)).astype(np.float64)
    Qd, Qh_, Qw = Pd, Ph, Pw                               # unit stride, "same" padding when pad = k // 2 (general: any pad)
    Q = rng.normal(size=(B, Qd, Qh_, Qw, Bc)).astype(np.float64)
    taps = [(z - pad[0], y - pad[1], x - pad[2]) for z in range(kd) for y in range(kh) for x in range(kw)]
    T = len(taps)
    ref = np.zeros((A, Bc, T))
    for t, (od, oh, ow) in enumerate(taps):
        for b in range(B):
            for d, h, w in itertools.product(range(Pd), range(Ph), range(Pw)):
                qd, qh, qw = d + od, h + oh, w + ow
                if 0 <= qd < Qd and 0 <= qh < Qh_ and 0 <= qw < Qw:
                    ref[:, :, t] += np.outer(P[b, d, h, w], Q[b, qd, qh, qw])
    flat = (Pd == 1 and kd == 1)
    TD, TH, TW = (1, 8, 16) if flat else (2, 8, 8)
    ROWH, LHM, CHS_P = (24 if TW == 16 else 16), TH + 2, 136
    CHS_Q = TD * LHM * ROWH + 8
    hmin, wmin = -pad[1], -pad[2]
    LH, LW = TH + (kh - 1), TW + (kw - 1)
    khw = kh * kw
    tgroups = kd if khw == 9 else 1
    tilesD, tilesH, tilesW = -(-Pd // TD), -(-Ph // TH), -(-Pw // TW)
    out = np.zeros((A, Bc, T))
    for a0, b0 in itertools.product(range(0, A, 32), range(0, Bc, 32)):
        for tg, b, tdi, thi, twi in itertools.product(range(tgroups), range(B), range(tilesD), range(tilesH), range(tilesW)):
            t0 = tg * khw
            od = taps[t0][0]
            p0d, p0h, p0w = tdi * TD, thi * TH, twi * TW
            Pl_ = np.full(32 * CHS_P, np.nan); Ql_ = np.full(32 * CHS_Q, np.nan)      # NaN = never written
            for it in range(64 * 8):
                c4, q0 = it & 7, (it >> 3) * 2
                pw, ph, pd = q0 % TW, (q0 // TW) % TH, q0 // (TW * TH)
                gd, gh, gw = p0d + pd, p0h + ph, p0w + pw
                for j in range(4):
                    ch = a0 + c4 * 4 + j
                    for e in range(2):
                        v = P[b, gd, gh, gw + e, ch] if (gd < Pd and gh < Ph and gw + e < Pw and ch < A) else 0.0
                        Pl_[(c4 * 4 + j) * CHS_P + q0 + e] = v
            npq = (LW + 1) >> 1
            for it in range(TD * LH * npq * 8):
                c4 = it & 7; r = it >> 3
                pr = r % npq; r //= npq
                lh, ld = r % LH, r // LH
                lw = pr * 2
                gd, gh, gw = p0d + od + ld, p0h + hmin + lh, p0w + wmin + lw
                off = (ld * LHM + lh) * ROWH + lw
                for j in range(4):
                    ch = b0 + c4 * 4 + j
                    for e in range(2):
                        ok = 0 <= gd < Qd and 0 <= gh < Qh_ and 0 <= gw + e < Qw and ch < Bc
                        Ql_[(c4 * 4 + j) * CHS_Q + off + e] = Q[b, gd, gh, gw + e, ch] if ok else 0.0
            acc = np.zeros((9, 32, 32))
            for wave in range(4):
                for i in range(2):
                    Afr = np.zeros((32, 2, 8)); Bfr = {}
                    for col, hh in itertools.product(range(32), range(2)):
                        hb = (wave * 2 + i) * 2 + hh
                        Afr[col, hh] = Pl_[col * CHS_P + hb * 8: col * CHS_P + hb * 8 + 8]
                        prow = hb if TW == 8 else (hb >> 1)
                        pd, ph = prow // TH, prow % TH
                        qoff = col * CHS_Q + (pd * LHM + ph) * ROWH + ((hb & 1) * 8 if TW == 16 else 0)
                        for dh in range(kh):
                            base = qoff + dh * ROWH
                            v10 = Ql_[base: base + 10]                    # b128 (8 halves) + b32 (2 halves)
                            for dw in range(kw):
                                Bfr[(col, hh, dh, dw)] = v10[dw: dw + 8]
                    for dh, dw in itertools.product(range(kh), range(kw)):
                        Bm = np.array([[Bfr[(col, hh, dh, dw)] for hh in range(2)] for col in range(32)])      # [32][2][8]
                        assert not np.isnan(Bm).any() and not np.isnan(Afr).any(), "fragment read an LDS slot nobody wrote"
                        acc[dh * 3 + dw] += np.einsum("ahe,bhe->ab", Afr, Bm)
            for dh, dw in itertools.product(range(kh), range(kw)):
                j = dh * kw + dw
                na, nb = min(32, A - a0), min(32, Bc - b0)
                out[a0:a0 + na, b0:b0 + nb, t0 + j] += acc[dh * 3 + dw][:na, :nb]
    err = np.abs(out - ref).max() / np.abs(ref).max()
    return err

cases = [dict(B=1, Pd=3, Ph=9, Pw=11, A=32, Bc=32, kd=3, kh=3, kw=3, pad=(1, 1, 1)),
         dict(B=2, Pd=4, Ph=8, Pw=8, A=40, Bc=33, kd=3, kh=3, kw=3, pad=(1, 1, 1)),
         dict(B=1, Pd=1, Ph=10, Pw=21, A=32, Bc=36, kd=1, kh=3, kw=3, pad=(0, 1, 1)),
         dict(B=1, Pd=1, Ph=8, Pw=16, A=16, Bc=64, kd=1, kh=1, kw=1, pad=(0, 0, 0)),
         dict(B=1, Pd=2, Ph=5, Pw=7, A=8, Bc=8, kd=1, kh=1, kw=1, pad=(0, 0, 0)),
         dict(B=1, Pd=2, Ph=9, Pw=9, A=32, Bc=32, kd=1, kh=3, kw=3, pad=(0, 1, 1))]
for c in cases:
    print(c, "max rel err", emulate(**c))
