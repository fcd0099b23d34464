"""CPU emulation of wgrad_f16x3_kernel's CLASS MODE index arithmetic (stride-2 convs and stride-2 transposed convs): parity classes, tap
groups by d delta, sub-lattice staging (tensor coordinate = 2 * (p0 + delta_min + l) + par), slot table, fragment windows -- against
torch autograd of F.conv3d / F.conv_transpose3d.  Mirrors the host tables of wgrad_impl and the kernel's addressing (csrc/wgrad.hip)."""
import itertools
import numpy as np
import torch
import torch.nn.functional as F

def host_tables(kd, kh, kw, pad, transposed):
    taps_od, taps_oh, taps_ow, tapid, c_t0, c_nt = [], [], [], [], [], []
    kk, pd3 = (kd, kh, kw), pad
    for c in range(8):
        par = ((c >> 2) & 1, (c >> 1) & 1, c & 1)
        idx, dele = [[], [], []], [[], [], []]
        for dim in range(3):
            for k in range(kk[dim]):
                off = k - pd3[dim]
                if off % 2 != par[dim]:
                    continue
                idx[dim].append(k); dele[dim].append((off - par[dim]) // 2)
        c_t0.append(len(tapid))
        for i, j, l in itertools.product(range(len(idx[0])), range(len(idx[1])), range(len(idx[2]))):
            taps_od.append(dele[0][i]); taps_oh.append(dele[1][j]); taps_ow.append(dele[2][l])
            tapid.append((idx[0][i] * kh + idx[1][j]) * kw + idx[2][l])
        c_nt.append(len(tapid) - c_t0[-1])
    groups = []
    for c in range(8):
        parh, parw = (c >> 1) & 1, c & 1
        i, e = c_t0[c], c_t0[c] + c_nt[c]
        while i < e:
            j = i
            while j < e and taps_od[j] == taps_od[i]:
                j += 1
            slot = [-1] * 9
            for t2 in range(i, j):
                dh, dw = taps_oh[t2] + parh, taps_ow[t2] + parw
                assert 0 <= dh <= 2 and 0 <= dw <= 2 and slot[dh * 3 + dw] < 0
                slot[dh * 3 + dw] = t2 - i
            groups.append(dict(t0=i, nt=j - i, par=c, od=taps_od[i], slot=slot))
            i = j
    return groups, tapid

def emulate(P, Q, A, Bc, kd, kh, kw, pad, transposed):
    """P [B,Pd,Ph,Pw,A], Q [B,Qd,Qh,Qw,Bc] (NDHWC); returns dW[A][Bc][T] by the kernel's class-mode arithmetic."""
    B, Pd, Ph, Pw, _ = P.shape
    _, Qd, Qh_, Qw, _ = Q.shape
    T = kd * kh * kw
    groups, tapid = host_tables(kd, kh, kw, pad, transposed)
    flat = (Pd == 1 and kd == 1)
    TD, TH, TW = (1, 8, 16) if flat else (2, 8, 8)
    ROWH, LHM, CHS_P = (24 if TW == 16 else 16), TH + 2, 136
    CHS_Q = TD * LHM * ROWH + 8
    LH, LW = TH + 1, TW + 1
    tilesD, tilesH, tilesW = -(-Pd // TD), -(-Ph // TH), -(-Pw // TW)
    out = np.zeros((A, Bc, T))
    for a0, b0 in itertools.product(range(0, A, 32), range(0, Bc, 32)):
        for g, b, tdi, thi, twi in itertools.product(groups, range(B), range(tilesD), range(tilesH), range(tilesW)):
            par = g["par"]; pard, parh, parw = (par >> 2) & 1, (par >> 1) & 1, par & 1
            od, ghmin, gwmin, qs = g["od"], -parh, -parw, 2
            p0d, p0h, p0w = tdi * TD, thi * TH, twi * TW
            Pl_ = np.full(32 * CHS_P, np.nan); Ql_ = np.full(32 * CHS_Q, np.nan)
            for it in range(64 * 8):
                c4, q0 = it & 7, (it >> 3) * 2
                pw, ph, pd = q0 % TW, (q0 // TW) % TH, q0 // (TW * TH)
                gd, gh, gw = p0d + pd, p0h + ph, p0w + pw
                for j in range(4):
                    ch = a0 + c4 * 4 + j
                    for e in range(2):
                        Pl_[(c4 * 4 + j) * CHS_P + q0 + e] = P[b, gd, gh, gw + e, ch] if (gd < Pd and gh < Ph and gw + e < Pw and ch < A) else 0.0
            npq = (LW + 1) >> 1
            for it in range(TD * LH * npq * 8):
                c4 = it & 7; r = it >> 3
                pr = r % npq; r //= npq
                lh, ld = r % LH, r // LH
                lw = pr * 2
                gd, gh, gw = qs * (p0d + od + ld) + pard, qs * (p0h + ghmin + lh) + parh, qs * (p0w + gwmin + lw) + parw
                off = (ld * LHM + lh) * ROWH + lw
                for j in range(4):
                    ch = b0 + c4 * 4 + j
                    for e in range(2):
                        ok = 0 <= gd < Qd and 0 <= gh < Qh_ and 0 <= gw + e * qs < Qw and ch < Bc
                        Ql_[(c4 * 4 + j) * CHS_Q + off + e] = Q[b, gd, gh, gw + e * qs, ch] if ok else 0.0
            acc = np.zeros((9, 32, 32))
            for wave, i in itertools.product(range(4), range(2)):
                Afr = np.zeros((32, 2, 8)); Bfr = {}
                for col, hh in itertools.product(range(32), range(2)):
                    hb = (wave * 2 + i) * 2 + hh
                    Afr[col, hh] = Pl_[col * CHS_P + hb * 8: col * CHS_P + hb * 8 + 8]
                    prow = hb if TW == 8 else (hb >> 1)
                    pd, ph = prow // TH, prow % TH
                    qoff = col * CHS_Q + (pd * LHM + ph) * ROWH + ((hb & 1) * 8 if TW == 16 else 0)
                    for dh in range(3):
                        v10 = Ql_[qoff + dh * ROWH: qoff + dh * ROWH + 10]
                        for dw in range(3):
                            Bfr[(col, hh, dh, dw)] = v10[dw: dw + 8]
                for dh, dw in itertools.product(range(3), range(3)):
                    if g["slot"][dh * 3 + dw] < 0:
                        continue
                    Bm = np.array([[Bfr[(col, hh, dh, dw)] for hh in range(2)] for col in range(32)])
                    assert not np.isnan(Bm).any() and not np.isnan(Afr).any(), "fragment read an LDS slot nobody wrote"
                    acc[dh * 3 + dw] += np.einsum("ahe,bhe->ab", Afr, Bm)
            for s_, j in enumerate(g["slot"]):
                if j >= 0:
                    na, nb = min(32, A - a0), min(32, Bc - b0)
                    out[a0:a0 + na, b0:b0 + nb, tapid[g["t0"] + j]] += acc[s_][:na, :nb]
    return out

def check(name, Ci, Co, k, pad, dims, transposed, opad=0, seed=0):
    torch.manual_seed(seed)
    B = 1
    x = torch.randn(B, Ci, *dims, dtype=torch.float64)
    if transposed:
        w = torch.randn(Ci, Co, *k, dtype=torch.float64, requires_grad=True)
        y = F.conv_transpose3d(x, w, None, 2, pad, opad)
    else:
        w = torch.randn(Co, Ci, *k, dtype=torch.float64, requires_grad=True)
        y = F.conv3d(x, w, None, 2, pad)
    dy = torch.randn_like(y)
    y.backward(dy)
    xn, dyn = x.permute(0, 2, 3, 4, 1).numpy(), dy.detach().permute(0, 2, 3, 4, 1).numpy()
    if transposed:      # P = x (a = ci), Q = dy (b = co), dW [Ci][Co][k]
        got = emulate(xn, dyn, Ci, Co, *k, pad, True)
    else:               # P = dy (a = co), Q = x (b = ci), dW [Co][Ci][k]
        got = emulate(dyn, xn, Co, Ci, *k, pad, False)
    want = w.grad.reshape(got.shape).numpy()
    print(f"{name:44s} max rel err {np.abs(got - want).max() / np.abs(want).max():.2e}")

check("conv3d 3x3x3 stride 2 pad 1, 8->8 @4x10x12", 8, 8, (3, 3, 3), (1, 1, 1), (4, 10, 12), False)
check("conv3d 3x3x3 stride 2 pad 1, 36->33 @6x8x18", 36, 33, (3, 3, 3), (1, 1, 1), (6, 8, 18), False)
check("deconv3d k3 pad 1 opad 1, 8->8 @3x5x7", 8, 8, (3, 3, 3), (1, 1, 1), (3, 5, 7), True, 1)
check("deconv3d k4 pad 1, 16->9 @3x6x9", 16, 9, (4, 4, 4), (1, 1, 1), (3, 6, 9), True, 0)
check("deconv2d k4 pad 1 (D = 1), 8->9 @1x9x20", 8, 9, (1, 4, 4), (0, 1, 1), (1, 9, 20), True, 0)
