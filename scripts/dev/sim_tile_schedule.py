"""Processor-sharing simulation of the blend kernels' tile schedule (input: gpurun_out/tile_times_{fwd,bwd}_N.npy written
by diag_tile_times.py).  1024 SIMDs, w wave slots each, a SIMD's instruction rate depends on how many waves it holds
(profiles/r02_occupancy_sensitivity.txt), waves are placed in dispatch order on the first free slot.  Compares dispatch
orders and tile splits against the measured makespan and the perfectly balanced bound."""
import sys

import numpy as np


def simulate(work, w, rate, S=1024):
    """work: per-unit instruction counts in dispatch order.  Returns (makespan, finish times)."""
    n = len(work)
    rem = np.zeros((S, w))          # remaining work per slot
    busy = np.zeros((S, w), bool)
    owner = -np.ones((S, w), int)
    nxt = 0
    # initial placement: round-robin over SIMDs, slot by slot
    for slot in range(w):
        for s in range(S):
            if nxt < n:
                rem[s, slot] = work[nxt]; busy[s, slot] = True; owner[s, slot] = nxt; nxt += 1
    t = 0.0
    fin = np.zeros(n)
    r = np.array([0.0] + [rate(k) for k in range(1, w + 1)])
    while busy.any():
        cnt = busy.sum(1)
        per_wave = np.where(cnt > 0, r[cnt] / np.maximum(cnt, 1), 0.0)  # instr per ns per wave
        tt = np.where(busy, rem / np.maximum(per_wave[:, None], 1e-30), np.inf)
        dt = tt.min()
        t += dt
        rem = np.where(busy, rem - dt * per_wave[:, None], rem)
        done = busy & (tt <= dt * (1 + 1e-12))
        for s, slot in zip(*np.nonzero(done)):
            fin[owner[s, slot]] = t
            if nxt < n:
                rem[s, slot] = work[nxt]; owner[s, slot] = nxt; nxt += 1
            else:
                busy[s, slot] = False
    return t, fin


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "bwd"
    d = np.load("gpurun_out/tile_times_%s_%s.npy" % (which, sys.argv[2] if len(sys.argv) > 2 else "14"))
    t0, t1, lst, walked = d[:4]
    npairs, nbody = walked, 2.17 * walked  # executed pairs ~ walked depth; 2.17 quadrant bodies per pair (measured)
    if which == "bwd":
        w, a, b, pro = 4, 63.0, 36.0, 400.0
        rate = lambda k: 1.0 / (214.0 + 384.0 / k)
    else:
        w, a, b, pro = 6, 22.0, 27.0, 300.0
        rate = lambda k: 1.0 / (79.5 + 303.0 / k)
    work = pro + a * npairs + b * nbody
    meas = t1.max()
    base, _ = simulate(work, w, rate)
    scale = meas / base  # calibrates instruction units -> us
    ideal = work.sum() / (1024 * rate(w)) * scale
    print("%s: measured makespan %.1f us; simulated (dispatch order as run) %.1f (scaled to match); perfectly balanced %.1f us" % (which, meas, base * scale, ideal))
    order = np.argsort(-work)
    m, _ = simulate(work[order], w, rate)
    print("  true LPT (exact work key):                      %.1f us" % (m * scale))
    rng = np.random.default_rng(0)
    m, _ = simulate(work[rng.permutation(len(work))], w, rate)
    print("  random order:                                   %.1f us" % (m * scale))
    for wv in ([3, 5] if which == "bwd" else [4, 5]):
        m, _ = simulate(work[order], wv, rate)
        print("  true LPT, %d slots per SIMD:                      %.1f us" % (wv, m * scale))
    for seg in (256, 128, 64):
        units = []
        for p_, b_ in zip(npairs, nbody):
            k = max(1, int(np.ceil(p_ / seg)))
            units += [pro + (a * p_ + b * b_) / k] * k
        units = np.array(units)
        m, _ = simulate(units[np.argsort(-units)], w, rate)
        print("  split into <= %3d-pair units (%5d units), LPT:   %.1f us   (ideal incl. prologues %.1f)" % (
            seg, len(units), m * scale, units.sum() / (1024 * rate(w)) * scale))


if __name__ == "__main__":
    main()
