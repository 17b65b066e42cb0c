#!/usr/bin/env python3
"""zir_model.py -- numerical model of the round-2 whole-tile kernels' reset handling (CPU, numpy).

resetState() (fsk.ts:175-188) zeroes the NCO phase and the I/Q low-pass state in the middle of the sample loop.  The
pipelined kernels keep the front end (AGC -> pre-filter -> mix -> I/Q low-pass) FREE-RUNNING -- it never sees a reset --
and the back end corrects its pair sums for a reset at input sample n0 by linearity:

    zero-state response after n0  =  e^{-j w n0} * ( free-running output  -  zero-input response of the state at n0 )

The zero-input response (ZIR) of the decimated pair sums q[m] = Z[2m] + Z[2m+1] obeys
    q[m+2] = (a1^2 - 2 a2) q[m+1] - a2^2 q[m]
and its first two values are obtained from two directly computed pairs.  The constant rotation only matters for the
first phase difference after the reset (lastPhase = 0 in the reset frame == w*n0 in the free frame).

Which state the ZIR is taken of matters in float32.  lag = 0 subtracts the response of the free-running state AT the
reset: exact algebra, but where the input has collapsed before the reset (the ringing after a frame that is followed by
digital silence) U and q agree to ~1e-8 and their float32 difference is rounding noise.  The kernels use lag = 16: a
zero-started direct instance produces the 16 + 2 decimated samples after the reset, the free-running filters are zeroed
16 decimated samples after it, and the ZIR that remains to be subtracted is that of the DIRECT instance's state there --
of the size of the wanted signal itself.  main() shows both on a noisy signal and on such a ringing tail.

This script checks, in float64 and in float32, that the scheme reproduces a sample-serial implementation with real
resets, and prints the constants the host derives (fsk_api.hip).  It is a design aid, not part of the product or tests.
"""
import math
import numpy as np


def butter_lp(cutoff, sr):
    c = math.tan(math.pi * (cutoff / (sr / 2)) / 2)
    den = 1 + math.sqrt(2) * c + c * c
    b0 = c * c / den
    return b0, (2 * c * c - 2) / den, (1 - math.sqrt(2) * c + c * c) / den


def serial_with_resets(y, w, b0, a1, a2, resets, dt):
    """reference order: mix with an NCO that restarts at a reset, DF-I biquads, pair sums; returns complex pair sums"""
    n = len(y)
    out = np.zeros(n // 2, dtype=np.complex128)
    x1 = x2 = y1 = y2 = 0j
    ph = 0.0
    acc = 0j
    for i in range(n):
        if i in resets:
            x1 = x2 = y1 = y2 = 0j
            ph = 0.0
        x = dt(y[i]) * complex(math.cos(ph), math.sin(ph))
        ph = math.fmod(ph + w, 2 * math.pi)
        o = b0 * x + 2 * b0 * x1 + b0 * x2 - a1 * y1 - a2 * y2
        x2, x1 = x1, x
        y2, y1 = y1, o
        acc += o
        if i & 1:
            out[i // 2] = acc / 2
            acc = 0j
    return out


def free_plus_zir(y, w, b0, a1, a2, resets, f=np.float64, lag=16):
    """the kernels' order: free-running front, ZIR-corrected back; all arithmetic rounded to dtype f.
    lag: decimated samples after a reset at which the free-running filters are zeroed (0: never, the first design)"""
    n = len(y)
    c1 = f(a1 * a1 - 2 * a2)
    c2 = f(a2 * a2)
    delta = f(1 + a1 + a2)
    a2f = f(a2)
    b0h = f(b0 / 2)
    cplx = np.complex64 if f is np.float32 else np.complex128

    def lp_step(st, x):  # velocity form, half scale (fsk_demod.hip lp32): st = [x1, x2, y, v]
        t = cplx(2) * st[0] + x + st[1]
        u = b0h * t - delta * st[2]
        st[3] = cplx(a2f * st[3] + u)
        st[2] = cplx(st[2] + st[3])
        st[1] = st[0]
        st[0] = x
        return st[2]

    F = [cplx(0)] * 4
    D = [cplx(0)] * 4
    qa = qb = cplx(0)
    direct_pairs = lag + 2
    dphase = direct_pairs
    q0 = cplx(0)
    rot = 0.0
    out = np.zeros(n // 2, dtype=np.complex128)
    frame_rot = np.zeros(n // 2)
    for m in range(n // 2):
        if 2 * m in resets:
            D = [cplx(0)] * 4
            dphase = 0
            rot = w * (2 * m)
        zs = [cplx(complex(math.cos(w * i), math.sin(w * i))) for i in (2 * m, 2 * m + 1)]
        xs = [cplx(f(y[2 * m + j]) * zs[j]) for j in (0, 1)]
        if lag and dphase == lag:
            F = [cplx(0)] * 4
        U = cplx(lp_step(F, xs[0]) + lp_step(F, xs[1]))
        if dphase < direct_pairs:
            Wd = cplx(lp_step(D, xs[0]) + lp_step(D, xs[1]))
            if dphase == lag:
                q0 = cplx(U - Wd)
            elif dphase == lag + 1:
                q1 = cplx(U - Wd)
                qa = cplx(c1 * q1 - c2 * q0)
                qb = cplx(c1 * qa - c2 * q1)
            dphase += 1
            wv = Wd
        else:
            wv = cplx(U - qa)
            qa, qb = qb, cplx(c1 * qb - c2 * qa)
        out[m] = complex(wv)
        frame_rot[m] = rot
    return out, frame_rot


def main():
    sr, baud = 48000.0, 1200.0
    b0, a1, a2 = butter_lp(baud, sr)
    w = 2 * math.pi * 1700.0 / sr
    rng = np.random.default_rng(1)
    n = 4000
    t = np.arange(n)
    y = 0.4 * np.sin(2 * math.pi * 1200 / sr * t) + 0.05 * rng.standard_normal(n)
    resets = {0, 500, 506, 1200, 3000}
    ref = serial_with_resets(y, w, b0, a1, a2, resets, float)
    print("lp b0 %.17g a1 %.17g a2 %.17g   c1 %.17g c2 %.17g" % (b0, a1, a2, a1 * a1 - 2 * a2, a2 * a2))
    for lag in (0, 16):
        for f in (np.float64, np.float32):
            got, rot = free_plus_zir(y, w, b0, a1, a2, resets, f, lag)
            got_rot = got * np.exp(-1j * rot)  # back into the reset frame
            peak = np.abs(ref).max()
            err = np.abs(got_rot - ref)
            amp_err = np.abs(np.abs(got) - np.abs(ref))
            rel_floor = amp_err / np.maximum(np.abs(ref), 0.01 * peak)
            print("noisy tone, lag %2d, %s: max |err| %.3e (peak %.3f)  max amp err vs max(ref,1%% peak) %.3e  plain rel amp err where ref>1e-3*peak %.3e"
                  % (lag, f.__name__, err.max(), peak, rel_floor.max(),
                     (amp_err / np.abs(ref))[np.abs(ref) > 1e-3 * peak].max()))
    # the case tools/soak.py found, in ITU-T V.21 numbers (300 baud: the I/Q low-pass, pole radius 0.973, rings longer than
    # the 800 Hz pre-filter, radius 0.949): a tone stops, and 'eod' resets the state 560 samples into the silence.  What
    # the reference's zero-started filters produce from there is ~1e-7 of what the free-running ones still hold.
    b3, a31, a32 = butter_lp(300.0, sr)
    w3 = 2 * math.pi * 1170.0 / sr
    yr = np.zeros(2400)
    yr[:800] = 0.4 * np.sin(2 * math.pi * 1070 / sr * np.arange(800))
    r = 0.949
    for i in range(800, 2400):   # pre-filter ringing: y[n] = 2 r cos(w) y[n-1] - r^2 y[n-2]
        yr[i] = 2 * r * math.cos(w3) * yr[i - 1] - r * r * yr[i - 2]
    rs = {0, 1360}
    ref = serial_with_resets(yr, w3, b3, a31, a32, rs, float)
    for lag in (0, 16):
        got, rot = free_plus_zir(yr, w3, b3, a31, a32, rs, np.float32, lag)
        free, _ = free_plus_zir(yr, w3, b3, a31, a32, {0}, np.float32, lag)
        got_rot = got * np.exp(-1j * rot)
        sel = slice(680 + 2, 680 + 120)
        rel = np.abs(got_rot[sel] - ref[sel]) / np.abs(ref[sel])
        print("ringing tail, lag %2d, float32: relative error of 118 pair sums after the reset: max %.3e  median %.3e   (|reference| / |free-running| there: %.1e)"
              % (lag, rel.max(), np.median(rel), np.median(np.abs(ref[sel]) / np.abs(free[sel]))))
    # conversion back to a full-rate zero-input state at an even time: (qa, qb) -> (Z[n-1], Z[n-1]-Z[n-2])
    def q_of(z1, z2):
        Z = [z2, z1]
        for _ in range(4):
            Z.append(-a1 * Z[-1] - a2 * Z[-2])
        return Z[2] + Z[3], Z[4] + Z[5]
    Lm = np.array([q_of(1, 0), q_of(0, 1)]).T   # columns: response to unit zeta1, zeta2
    Linv = np.linalg.inv(Lm)
    print("q = L (zeta1, zeta2):", Lm.tolist(), "cond %.3g" % np.linalg.cond(Lm))
    # rows: y_Z = zeta1, v_Z = zeta1 - zeta2
    conv = np.array([Linv[0], Linv[0] - Linv[1]])
    print("(y_Z, v_Z) = conv (qa, qb):", conv.tolist())


if __name__ == "__main__":
    main()
