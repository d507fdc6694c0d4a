"""Offline model of the LDS bank behaviour of render_bwd_tile_kernel's deposits (r03 study, not part of the product).

For a camera of the synthetic set, sample (tile, sample index) pairs, place the 64 lanes' 2x2x2 footprints in the sheared window
exactly like the kernel (reference lane 27, ring of 6, 8x8 lateral, per-layer rotation, per-lane corner / channel rotation) and
count, per ds_add_f64 wave instruction, a conflict cost under simple bank models.  Used to compare lane -> pixel maps and LDS
index maps across views before building them (profiles/r03_ab_orientation.txt has the hardware numbers the model is held against).

  python tools/sim/lds_conflicts.py [cams...]
"""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vox-e_amd"))
from voxe_hip.workload import synth_pose_angles, RADIUS, NEAR, FAR, focal_for  # noqa: E402
from thre3d_atom.utils.imaging_utils import pose_spherical  # noqa: E402

G, HW, S = 160, 400, 256
KRING, KL, ROT, PAD = 6, 8, 25, 6
KPLANE = KRING * KL * KL + PAD


def rays_of(cam):
    pose = pose_spherical(*synth_pose_angles(cam, 100), RADIUS)
    R = np.asarray(pose.rotation, dtype=np.float64).reshape(3, 3)
    t = np.asarray(pose.translation, dtype=np.float64).reshape(3)
    f = focal_for(HW)
    j, i = np.meshgrid(np.arange(HW) + 0.5, np.arange(HW) + 0.5, indexing="xy")   # j: x (column), i: y (row)
    dirs = np.stack([(j - HW / 2) / f, -(i - HW / 2) / f, -np.ones_like(j)], -1)
    d = dirs @ R.T
    return t, d        # d[row, col]


def lane_pixels(mapping):
    lane = np.arange(64)
    if mapping == "rows":
        return lane >> 3, lane & 7           # (row, col) inside the tile
    if mapping == "cols":
        return lane & 7, lane >> 3
    raise ValueError(mapping)


def cost_of(idx, model):
    """idx: [64] LDS double indices of one wave instruction -> cycles-ish"""
    bank = (2 * idx) % 64
    if model == "half32":      # two passes of 32 lanes; a pass costs the largest number of lanes on one bank pair
        c = 0
        for h in (slice(0, 32), slice(32, 64)):
            _, cnt = np.unique(bank[h], return_counts=True)
            c += cnt.max()
        return c
    if model == "full64":
        _, cnt = np.unique(bank, return_counts=True)
        return cnt.max()
    if model == "distinct":    # same address combined for free, distinct addresses on one bank serialise
        c = 0
        for h in (slice(0, 32), slice(32, 64)):
            u = np.unique(idx[h])
            _, cnt = np.unique((2 * u) % 64, return_counts=True)
            c += cnt.max()
        return c
    raise ValueError(model)


def simulate(cam, mapping="rows", lanerot=None, crot=None, posfn=None, ntiles=60, seed=0, models=("half32", "full64", "distinct")):
    rng = np.random.default_rng(seed)
    o, d = rays_of(cam)
    prow, pcol = lane_pixels(mapping)
    lane = np.arange(64)
    lanerot = lanerot or (lambda l: ((l & 7) + 3 * (l >> 3)) & 7)
    crot = crot or (lambda l: l & 3)
    posfn = posfn or (lambda a, b, sl: (a * 8 + b + ROT * sl) & 63)
    rot = lanerot(lane)
    r0, r1, r2 = rot & 1, (rot >> 1) & 1, (rot >> 2) & 1
    cr = crot(lane)
    step = (FAR - NEAR) / (S - 1)
    tot = {m: 0.0 for m in models}
    n_instr = 0
    fallback = 0
    for _ in range(ntiles):
        ty, tx = rng.integers(5, HW // 8 - 5, 2)
        dd = d[ty * 8 + prow, tx * 8 + pcol]                # [64,3]
        for k in rng.integers(40, 216, 6):
            z = NEAR + step * (k + rng.random(64) - 0.5)   # jittered depths
            p = o[None, :] + dd * z[:, None]
            U = (p * (2.0 / 3.0) + 1.0) * (G / 2) - 0.5
            inside = np.all((p > -1.5) & (p < 1.5), axis=1)
            if inside.sum() < 48:
                continue
            i0 = np.floor(U).astype(int)
            ref = 27
            U0 = (o * (2.0 / 3.0) + 1.0) * (G / 2) - 0.5
            DU = dd[ref] * (2.0 / 3.0) * (G / 2)
            m = int(np.argmax(np.abs(DU)))
            u_ax = 1 if m == 0 else 0
            v_ax = 1 if m == 2 else 2
            sgn = -1 if DU[m] < 0 else 1
            Bu, Bv = DU[u_ax] / DU[m], DU[v_ax] / DU[m]
            Au, Av = U0[u_ax] - Bu * U0[m], U0[v_ax] - Bv * U0[m]
            off_u = lambda im: np.floor(Au + Bu * im).astype(int) - 3
            off_v = lambda im: np.floor(Av + Bv * im).astype(int) - 3
            pm, pu, pv = i0[:, m], i0[:, u_ax], i0[:, v_ax]
            key_lo = np.where(sgn > 0, pm, -(pm + 1))
            base = key_lo[inside].min()
            idxs = np.zeros((8, 64), dtype=int)
            ok = inside.copy()
            for dm in (0, 1):
                im = pm + dm
                key = sgn * im
                a0, b0 = pu - off_u(im), pv - off_v(im)
                ok &= (key - base >= 0) & (key - base < KRING) & (a0 >= 0) & (a0 < 7) & (b0 >= 0) & (b0 < 7)
            fallback += int((inside & ~ok).sum())
            for cc in range(8):
                bm, bu, bv = cc & 1, (cc >> 1) & 1, cc >> 2
                dm, du, dv = bm ^ r0, bu ^ r1, bv ^ r2
                im = pm + dm
                key = sgn * im
                sl = np.mod(key, KRING)
                a = pu - off_u(im) + du
                b = pv - off_v(im) + dv
                idxs[cc] = sl * 64 + posfn(a, b, sl)
            for cc in range(8):
                for j in range(4):
                    ch = (j + cr) & 3
                    idx = ch * KPLANE + idxs[cc]
                    act = ok
                    if act.sum() < 32:
                        continue
                    full = np.where(act, idx, -1 - lane * 2)   # inactive lanes: private dummy addresses (no conflicts)
                    for mname in models:
                        tot[mname] += cost_of(full, mname)
                    n_instr += 1
    return {m: round(tot[m] / max(n_instr, 1), 3) for m in models}, n_instr, fallback


if __name__ == "__main__":
    cams = [int(x) for x in sys.argv[1:]] or [3, 0, 50, 77, 26, 20, 13, 40, 90, 12]
    for cam in cams:
        r1 = simulate(cam, "rows")
        r2 = simulate(cam, "cols")
        print(f"cam {cam:3d}  rows {r1[0]}  cols {r2[0]}   (instr {r1[1]}, fallback lanes {r1[2]})")
