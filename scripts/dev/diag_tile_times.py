"""Load balance of blend_bwd: every wave (= tile) stamps its start / end (100 MHz wall clock) into a buffer handed over
through FSGS_DBG_TILE_TIMES.  Prints the makespan, the mean number of resident waves, the occupancy over time and how well
the LPT key (list length) predicts a tile's duration compared with the depth actually walked (max n_contrib).
    gpurun -- 'FSGS_DIAG=1 python free-surgs_amd/build.py && FSGS_LIB_PATH=free-surgs_amd/fsgs_amd/lib/diag/libfsgs_hip.diag.so python scripts/dev/diag_tile_times.py [--fwd]'
(the stamps are a diagnostics hook: only a library built with FSGS_DIAG=1 looks at the environment variable)"""
import json
import os
os.environ.setdefault("FSGS_BLEND_VARIANT", "one")  # the hooks live in the one-wave flavour of the blend kernels (round 5: the forward defaults to four waves per tile)
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import bench  # noqa: E402
os.makedirs('gpurun_out', exist_ok=True)


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    torch.manual_seed(0)
    np.random.seed(0)
    ntiles = 80 * 64
    nwaves = ntiles
    buf = torch.zeros((nwaves * 6,), dtype=torch.int64, device=dev)  # kDiagStampWords
    os.environ["FSGS_DBG_TILE_TIMES_FWD" if "--fwd" in sys.argv else "FSGS_DBG_TILE_TIMES"] = str(buf.data_ptr())
    slots = 6144 if "--fwd" in sys.argv else 5120
    from fsgs_amd import _lib
    from fsgs_amd.fast_step import FastStepper

    _lib.load()
    pc, poses, frames, cam, sc = bench.build_problem("C2", dev, 0, 1)
    st = FastStepper(pc, poses, frames)
    for it in range(12):
        st.mapping_step([it % len(frames.colors)])
    torch.cuda.synchronize()
    for it in range(12, 15):
        buf.zero_()
        st.mapping_step([it % len(frames.colors)])
        torch.cuda.synchronize()
        d = buf.cpu().numpy().reshape(nwaves, 6)
        t0, t1 = d[:, 0].astype(np.float64) * 0.01, d[:, 1].astype(np.float64) * 0.01  # us
        ok = d[:, 1] > 0
        if not ok.any():
            sys.exit("no stamps: needs the diagnostics flavour (FSGS_DIAG=1 python free-surgs_amd/build.py; FSGS_LIB_PATH=free-surgs_amd/fsgs_amd/lib/diag/libfsgs_hip.diag.so)")
        lst = (d[:, 3] >> 32).astype(np.float64)
        walked = (d[:, 3] & 0xFFFFFFFF).astype(np.float64)
        start, end = t0[ok].min(), t1[ok].max()
        dur = (t1 - t0)[ok]
        print("step %d: %d waves, makespan %.1f us, busy %.0f wave-us -> mean resident waves %.0f of %d (%.1f %%)" % (
            it, ok.sum(), end - start, dur.sum(), dur.sum() / (end - start), slots, 100 * dur.sum() / (end - start) / slots))
        # round 6: the shader clock the tiles' waves ran at = delta s_memtime (shader cycles) / delta s_memrealtime (100 MHz)
        cyc = (d[:, 5] - d[:, 4]).astype(np.float64)[ok]
        mhz = cyc / np.maximum(dur, 1e-9)
        print("  shader clock while the kernel ran: %.0f MHz (sum of cycles / sum of wall time over the waves; per wave p10 %.0f "
              "p50 %.0f p90 %.0f MHz)" % (cyc.sum() / dur.sum(), np.percentile(mhz, 10), np.percentile(mhz, 50), np.percentile(mhz, 90)))
        print(json.dumps({"kernel": "blend_fwd" if "--fwd" in sys.argv else "blend_bwd", "step": it, "shader_clock_mhz": cyc.sum() / dur.sum(),
                          "makespan_us": end - start, "waves": int(ok.sum())}))
        print("  tile duration us: mean %.1f  median %.1f  p90 %.1f  max %.1f;  list length mean %.0f max %.0f;  walked mean %.0f max %.0f" % (
            dur.mean(), np.median(dur), np.percentile(dur, 90), dur.max(), lst[ok].mean(), lst[ok].max(), walked[ok].mean(), walked[ok].max()))
        print("  corr(duration, list length) %.3f   corr(duration, walked depth) %.3f" % (
            np.corrcoef(dur, lst[ok])[0, 1], np.corrcoef(dur, walked[ok])[0, 1]))
        bodies = (d[:, 2] & 0xFFFFFFFF).astype(np.float64)  # quadrant bodies executed, pairs not skipped altogether
        pairs = (d[:, 2] >> 32).astype(np.float64)
        np.save("gpurun_out/tile_times_%s_%d.npy" % ("fwd" if "--fwd" in sys.argv else "bwd", it),
                np.stack([t0[ok] - start, t1[ok] - start, lst[ok], walked[ok], bodies[ok], pairs[ok]]))
        if ok.all() and nwaves % 1024 == 0 and bodies.sum() > 0:
            fin_ = (t1 - start).reshape(-1, 1024).max(axis=0)
            for nm, v in (("bodies", bodies), ("pairs", pairs), ("35 bodies + 63 pairs", 35 * bodies + 63 * pairs)):
                sv = v.reshape(-1, 1024).sum(axis=0)
                print("  per SIMD: sum of %s mean %.0f min %.0f max %.0f (max / mean %.3f), corr with the finish time %.3f" % (
                    nm, sv.mean(), sv.min(), sv.max(), sv.max() / sv.mean(), np.corrcoef(fin_, sv)[0, 1]))
            print("  per tile: bodies / list length mean %.2f p10 %.2f p90 %.2f; corr(bodies, list length) %.3f" % (
                (bodies / np.maximum(lst, 1)).mean(), np.percentile(bodies / np.maximum(lst, 1), 10),
                np.percentile(bodies / np.maximum(lst, 1), 90), np.corrcoef(bodies, lst)[0, 1]))
        for lo_, hi_ in ((0, 100), (100, 150), (150, 200), (200, 250), (250, 300), (300, 1000)):
            sel = (lst[ok] >= lo_) & (lst[ok] < hi_)
            if sel.any():
                print("  list length %3d..%3d: %4d tiles, duration mean %.1f us, start mean %.1f us" % (
                    lo_, hi_, sel.sum(), dur[sel].mean(), (t0[ok][sel] - start).mean()))
        # per SIMD: workgroup b of these one-wave launches shares its SIMD with b + 1024, b + 2048, ... (profiles/r03_dispatch_map.txt)
        if ok.all() and nwaves % 1024 == 0:
            fin = (t1 - start).reshape(-1, 1024).max(axis=0)           # when each SIMD ran dry
            work = lst.reshape(-1, 1024).sum(axis=0)                   # its tiles' list lengths
            wk2 = walked.reshape(-1, 1024).sum(axis=0)
            print("  per SIMD: finish time mean %.1f  p10 %.1f  p50 %.1f  p90 %.1f  max %.1f us (makespan / mean %.3f); sum of its list "
                  "lengths mean %.0f min %.0f max %.0f (max / mean %.3f); corr(finish, sum of lengths) %.3f, corr(finish, sum walked) %.3f" % (
                      fin.mean(), np.percentile(fin, 10), np.percentile(fin, 50), np.percentile(fin, 90), fin.max(), fin.max() / fin.mean(),
                      work.mean(), work.min(), work.max(), work.max() / work.mean(), np.corrcoef(fin, work)[0, 1], np.corrcoef(fin, wk2)[0, 1]))
            cu = fin.reshape(-1)  # by SIMD id = b % 1024; XCD = b % 8
            print("  per XCD finish (mean of its SIMDs): %s" % " ".join("%.0f" % cu[x::8].mean() for x in range(8)))
        nb = 16
        edges = np.linspace(start, end, nb + 1)
        occ = []
        for b in range(nb):
            lo, hi = edges[b], edges[b + 1]
            occ.append((np.clip(np.minimum(t1[ok], hi) - np.maximum(t0[ok], lo), 0, None)).sum() / (hi - lo))
        print("  resident waves over time (%d bins): %s" % (nb, " ".join("%.0f" % o for o in occ)))
        # start time of the k-th dispatched wave: when did the dispatcher run out of queued tiles?
        order = np.argsort(t0[ok])
        print("  last wave started at %.1f us of %.1f; waves started after 50 %% of the makespan: %d" % (
            t0[ok][order[-1]] - start, end - start, int((t0[ok] - start > 0.5 * (end - start)).sum())))


if __name__ == "__main__":
    main()
