"""Which dispatch order would the blend kernels like?  (FSGS_DIAG=1 build: FSGS_DBG_ORDER_FWD / _BWD inject an order made
here on the host.)  One frame of C2, the cloud frozen (lr 0): the list lengths are read back from the state, several orders
are formed from them and each is timed over 30 forward + backward passes with the library's own HIP events.
    gpurun -- 'FSGS_DIAG=1 python free-surgs_amd/build.py && FSGS_LIB_PATH=free-surgs_amd/fsgs_amd/lib/diag/libfsgs_hip.diag.so python scripts/dev/order_experiment.py'"""
import os
os.environ.setdefault("FSGS_BLEND_VARIANT", "one")  # the hooks live in the one-wave flavour of the blend kernels (round 5: the forward defaults to four waves per tile)
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import bench  # noqa: E402

NS = 1024  # SIMDs: block b shares its SIMD with b + 1024, ...


def fold(rank_desc, rev_mask, nrounds):
    """tiles by descending key -> positions, round k backwards when bit k of rev_mask"""
    out = np.full(nrounds * NS, 0xFFFFFFFF, np.uint32)
    n = len(rank_desc)
    for k in range(nrounds):
        seg = rank_desc[k * NS:(k + 1) * NS]
        if len(seg) == 0:
            break
        pos = np.arange(len(seg))
        if (rev_mask >> k) & 1:
            pos = len(seg) - 1 - pos
        out[k * NS + pos] = seg
    return out


def greedy_rounds(key, cap_rounds, load_fn=None):
    """round by round: the round's tiles (descending) go to the SIMDs in ascending order of their load so far"""
    idx = np.argsort(-key, kind="stable")
    load = np.zeros(NS)
    out = np.full(cap_rounds * NS, 0xFFFFFFFF, np.uint32)
    for k in range((len(idx) + NS - 1) // NS):
        seg = idx[k * NS:(k + 1) * NS]
        simd = np.argsort(load, kind="stable")[:len(seg)]
        out[k * NS + simd] = seg
        load[simd] += key[seg]
    return out


def lpt_capacity(key, cap):
    """LPT bin packing: every tile (descending) to the least loaded SIMD that still has a free slot; holes elsewhere"""
    import heapq
    idx = np.argsort(-key, kind="stable")
    heap = [(0.0, s) for s in range(NS)]
    cnt = np.zeros(NS, int)
    out = np.full(cap * NS, 0xFFFFFFFF, np.uint32)
    for t in idx:
        while True:
            l, s = heapq.heappop(heap)
            if cnt[s] < cap:
                break
        out[cnt[s] * NS + s] = t
        cnt[s] += 1
        heapq.heappush(heap, (l + key[t], s))
    return out


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    torch.manual_seed(0)
    np.random.seed(0)
    ntiles = 80 * 64
    obuf_f = torch.full((8 * NS,), -1, dtype=torch.int32, device=dev)  # 0xFFFFFFFF = a hole
    obuf_b = obuf_f.clone()
    # the hook reads the pointer once (static): the buffers stay, their contents change; grid = 6 rounds for both
    os.environ["FSGS_DBG_ORDER_FWD"] = str(obuf_f.data_ptr()); os.environ["FSGS_DBG_ORDER_FWD_N"] = str(6 * NS)
    os.environ["FSGS_DBG_ORDER_BWD"] = str(obuf_b.data_ptr()); os.environ["FSGS_DBG_ORDER_BWD_N"] = str(6 * NS)
    from fsgs_amd import _lib
    from fsgs_amd.fast_step import FastStepper

    lib = _lib.load()
    pc, poses, frames, cam, sc = bench.build_problem("C2", dev, 0, 1)
    for g in pc.optimizer.param_groups:
        g["lr"] = 0.0  # the cloud stands still: the same lists every pass
    st = FastStepper(pc, poses, frames)

    def put(buf, order):
        o = np.full(8 * NS, 0xFFFFFFFF, np.uint32)
        o[:len(order)] = order
        buf.copy_(torch.from_numpy(o.view(np.int32)).to(dev))

    ident = np.arange(ntiles, dtype=np.uint32)
    put(obuf_f, ident); put(obuf_b, ident)
    st.mapping_step([0])
    torch.cuda.synchronize()
    # list lengths from the state: ranges int2[tiles] at layout offset [3]
    import ctypes as C
    W, H, P = 1280, 1024, pc.num_points
    _, state, _, cap, _ = st._color_src
    off = (C.c_size_t * 9)()
    _lib.check(lib.fsgs_render_state_layout(P, W, H, int(cap), off), "fsgs_render_state_layout")
    rng = state[off[3]:off[3] + 8 * ntiles].view(torch.int32).reshape(ntiles, 2).cpu().numpy()
    length = (rng[:, 1] - rng[:, 0]).astype(np.float64)
    print("list lengths: mean %.0f max %.0f, R %d" % (length.mean(), length.max(), length.sum()))
    desc = np.argsort(-length, kind="stable").astype(np.uint32)

    def measure(tag, of, ob, reps=30):
        put(obuf_f, of); put(obuf_b, ob)
        for _ in range(3):
            st.mapping_step([0])
        torch.cuda.synchronize()
        _lib.profile_enable(["blend_fwd", "blend_bwd"])
        for _ in range(reps):
            st.mapping_step([0])
        torch.cuda.synchronize()
        pr = _lib.profile_read()
        _lib.profile_enable([])
        print("%-58s blend_fwd %.1f us  blend_bwd %.1f us" % (tag, 1e3 * pr["blend_fwd"][0] / pr["blend_fwd"][1],
                                                              1e3 * pr["blend_bwd"][0] / pr["blend_bwd"][1]))

    straight = fold(desc, 0, 5)
    alt = fold(desc, 0b01010, 5)
    measure("longest first, straight (round 2)", straight, straight)
    measure("alternating fold", alt, alt)
    for mask in (0b10000, 0b10010, 0b01111, 0b00001, 0b10101, 0b11010):
        o = fold(desc, mask, 5)
        measure("fold mask %s (bit k = round k backwards)" % format(mask, "05b"), o, o)
    g5 = greedy_rounds(length, 5)
    measure("greedy rounds on length (loads sorted every round)", g5, g5)
    # forward-specific: the longest tiles with fewer companions (6 slots per SIMD in the forward)
    for cap_ in (6,):
        o = lpt_capacity(length, cap_)
        measure("LPT packing on length, <= %d per SIMD (fwd), greedy rounds (bwd)" % cap_, o, g5)
    for p in (1.5, 2.0, 3.0):
        o = lpt_capacity(length ** p, 6)
        measure("LPT packing on length^%.1f, <= 6 per SIMD (fwd)" % p, o, g5)
    # concave key: the forward's cost saturates with the list length
    for sat in (150, 200, 250):
        o = greedy_rounds(np.minimum(length, sat) + 0.2 * length, 5)
        measure("greedy rounds on min(length, %d) + 0.2 length (fwd)" % sat, o, g5)


if __name__ == "__main__":
    main()
