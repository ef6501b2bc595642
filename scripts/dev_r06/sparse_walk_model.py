"""Host-side model of the wave-uniform sparse z-walk (round 6): how many plane steps of k_zfwd / k_zbwd are dead
under different skip granularities, on the bench's own clouds (numpy only, no GPU).
  python scripts/dev_r06/sparse_walk_model.py [cfg ...]"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import dpc_amd
from oracle import dpc_oracle_np as onp


def marks_for(cfg_id, B=None, K=None, N=None, D=None):
    c = dict(dpc_amd.synthetic.CONFIGS.get(cfg_id, {}))
    if cfg_id == 3:
        c = dict(B=32, N=8000, D=64, K=21, sigma=3.0)
    if B: c["B"] = B
    if K: c["K"] = K
    inp = dpc_amd.synthetic.make_inputs(c["B"], c["N"], dpc_amd.synthetic.SEED0 + cfg_id)
    tr = onp.transform_fwd(inp["pc"].astype(np.float64), inp["pose"].astype(np.float64))
    tr = (tr[0] if isinstance(tr, tuple) else tr).astype(np.float32)
    D = c["D"]; H = c["K"] // 2
    valid = np.all((tr >= -0.5) & (tr <= 0.5), axis=-1)
    cell = lambda i: np.floor((tr[..., i] + np.float32(0.5)) * np.float32(D - 1)).astype(np.int64).clip(0, D - 1)
    iz, iy, ix = cell(0), cell(1), cell(2)
    out = []
    for b in range(tr.shape[0]):
        v = valid[b]
        z0, y0, x0 = iz[b][v], iy[b][v], ix[b][v]
        mark = np.zeros((D + 1, D, D // 32), dtype=bool)
        c_lo, c_hi = np.maximum(x0 - H, 0) >> 5, np.minimum(x0 + 1 + H, D - 1) >> 5
        for dy in range(-H, H + 2):
            y = y0 + dy
            ok = (y >= 0) & (y < D)
            for dz in (0, 1):
                for cc in (c_lo, c_hi):
                    mark[z0[ok] + dz, y[ok], cc[ok]] = True
        out.append(mark[:D])
    return c, np.stack(out)      # [B, Dz, D, D/32]


def wave_masks(mark, D, rows=1):
    B, Dz = mark.shape[:2]
    if rows > 1:      # a wave = (4 / rows) chunks x `rows` rows
        cpw = 4 // rows
        m = mark.reshape(B, Dz, D // rows, rows, (D // 32) // cpw, cpw).any(-1).any(3)
        m = m.reshape(B, Dz, -1)
    elif D >= 128:      # a wave = 128 rays = 4 chunks of one row
        m = mark.reshape(B, Dz, D, D // 128, 4).any(-1)          # [B,Dz,D,waves/row]
        m = m.reshape(B, Dz, -1)
    else:             # a wave = 128 / D rows
        r = 128 // D
        m = mark.any(-1).reshape(B, Dz, D // r, r).any(-1)
    return np.moveaxis(m, 1, -1).reshape(-1, Dz)     # [waves, Dz]


def dead_steps(wm, lookback, Dz, T, lag=0):
    """step t (0..T-1) pushes plane t; dead iff no marked plane in [t - lookback, t]"""
    W = wm.shape[0]
    pad = np.zeros((W, T + lookback), dtype=bool)
    pad[:, lookback:lookback + Dz] = wm
    cs = np.concatenate([np.zeros((W, 1), int), np.cumsum(pad, 1)], 1)
    t = np.arange(T)
    cnt = cs[:, t + lookback + 1] - cs[:, t]
    return cnt == 0     # [W, T]


def report(cfg_id, **kw):
    c, mark = marks_for(cfg_id, **kw)
    for rows in ((1, 2, 4) if c["D"] >= 128 else (1,)):
        report1(cfg_id, c, mark, rows)


def report1(cfg_id, c, mark, rows):
    D, K = c["D"], c["K"]; Dz = D; h = K // 2
    wm = wave_masks(mark, D, rows)
    print("rows/wave %d:" % rows)
    print("cfg%d D=%d K=%d: chunks marked %.3f, wave-planes marked %.3f, waves all-empty %.3f" %
          (cfg_id, D, K, mark.mean(), wm.mean(), (~wm.any(1)).mean()))
    G = K if K >= 4 else 2 * K
    for name, T, look in (("zfwd", Dz + h, K), ("zbwd", Dz + 2 * h, 2 * K - 1)):
        d = dead_steps(wm, look, Dz, T)
        line = "  %s: plane-level dead %.3f" % (name, d.mean())
        for S in (2, 3, 4, 6, G):
            nb = (T + S - 1) // S
            dd = np.ones((d.shape[0], nb * S), bool); dd[:, :T] = d
            blk = dd.reshape(d.shape[0], nb, S).all(-1)
            line += " | S=%d: %.3f" % (S, (blk.sum() * S) / (d.shape[0] * T))
        print(line)
    # staged rules for zbwd: (A) input plane nonzero -> fwd FMAs; (B) G2[t-h] nonzero -> full DRC + adjoint FMAs; (C) output nonzero / store
    T = Dz + 2 * h
    inp_dead = dead_steps(wm, 0, Dz, T)
    g2_dead = dead_steps(wm, 2 * h, Dz, T)
    print("  zbwd staged: input-plane dead %.3f, G2 dead %.3f (both K=%d)" % (inp_dead.mean(), g2_dead.mean(), K))


if __name__ == "__main__":
    for a in (sys.argv[1:] or ["2", "5", "3"]):
        report(int(a), **({"B": 4} if int(a) == 5 else {}))
    report(3, K=9)


def balance_report(cfg_id, B=None):
    """per-SIMD work under the balanced 1024-thread work-groups (zray_tile_balanced), cost of a dead group = 0.13 of a dense one"""
    c, mark = marks_for(cfg_id, B=B)
    D, K = c["D"], c["K"]; Dz = D; h = K // 2; G = K
    NB = mark.shape[0]
    NC, RG = D // 32, D // 4
    tile_any = mark.reshape(NB, Dz, RG, 4, NC).any(3)            # [B, Dz, RG, NC]
    for name, T, look in (("zfwd", Dz + h, K), ("zbwd", Dz + 2 * h, 2 * K - 1)):
        ng = (T + G - 1) // G
        cost = np.zeros((NB, RG, NC))
        for g in range(ng):
            lo, hi = max(g * G - look, 0), min(g * G + G - 1, Dz - 1)
            dead = ~tile_any[:, lo:hi + 1].any(1) if hi >= lo else np.ones((NB, RG, NC), bool)
            cost += np.where(dead, 0.13, 1.0)
        cost /= ng
        mul2 = [0, 2, 3, 1]
        q = max(NC // 4, 1)
        nwg = (RG // 4) * q if NC >= 4 else RG // 8
        res = {}
        for hyp in ("w%4", "w/4"):
            simd = np.zeros((NB, nwg, 4))
            for g in range(nwg):
                for w in range(16):
                    s, j = w & 3, w >> 2
                    a, b = s ^ j, mul2[s] ^ j
                    if NC >= 4:
                        rg, cc = g // q + a * (RG // 4), g % q + q * b
                    else:
                        rg, cc = g + a * (RG // 4) + (b >> 1) * (RG // 8), b & 1
                    simd[:, g, s if hyp == "w%4" else j] += cost[:, rg, cc]
            res[hyp] = simd / 4
        # plain tile order, 4-wave work-groups: wave v -> chunk v % NC of row group v / NC; no control over which 4 WGs share a SIMD: report the per-wave spread
        print("cfg%d %s: mean cost %.3f | balanced WGs: per-SIMD mean %.3f max %.3f (w%%4), max %.3f (w/4); per-CU max %.3f | per-wave max %.3f" % (
            cfg_id, name, cost.mean(), res["w%4"].mean(), res["w%4"].max(), res["w/4"].max(), res["w%4"].mean(-1).max(), cost.max()))


if __name__ == "__main__" and os.environ.get("BALANCE"):
    balance_report(2)
    balance_report(5, B=4)


def balance_mixed(cfg_id, B=None):
    c, mark = marks_for(cfg_id, B=B)
    D, K = c["D"], c["K"]; Dz = D; h = K // 2; G = K
    NB = mark.shape[0]
    NC, RG = D // 32, D // 4
    tile_any = mark.reshape(NB, Dz, RG, 4, NC).any(3)
    for name, T, look in (("zfwd", Dz + h, K), ("zbwd", Dz + 2 * h, 2 * K - 1)):
        ng = (T + G - 1) // G
        cost = np.zeros((NB, RG, NC))
        for g in range(ng):
            lo, hi = max(g * G - look, 0), min(g * G + G - 1, Dz - 1)
            dead = ~tile_any[:, lo:hi + 1].any(1) if hi >= lo else np.ones((NB, RG, NC), bool)
            cost += np.where(dead, 0.13, 1.0)
        cost /= ng
        mul2 = [0, 2, 3, 1]
        q = max(NC // 4, 1)
        nwg = (RG // 4) * q if NC >= 4 else RG // 8
        for vs in (0, NB // 4, 1):        # view stride between the 4 waves of a SIMD
            simd = np.zeros((NB, nwg, 4))
            for by in range(NB):
                for g in range(nwg):
                    for w in range(16):
                        s, j = w & 3, w >> 2
                        a, b = s ^ j, mul2[s] ^ j
                        if NC >= 4:
                            rg, cc = g // q + a * (RG // 4), g % q + q * b
                        else:
                            rg, cc = g + a * (RG // 4) + (b >> 1) * (RG // 8), b & 1
                        simd[by, g, s] += cost[(by + j * vs) % NB, rg, cc]
            print("cfg%d %s view stride %d: per-SIMD mean %.3f max %.3f  p99 %.3f | per-CU max %.3f" % (cfg_id, name, vs, simd.mean() / 4, simd.max() / 4,
                  np.percentile(simd, 99) / 4, simd.mean(-1).max() / 4))


if __name__ == "__main__" and os.environ.get("BALANCE2"):
    balance_mixed(2)
    balance_mixed(5, B=8)


def yblur_blocks(cfg_id, B=None):
    """k_splat_xy's y-blur by column block: of the (plane, strip) tiles that hold anything, which share of their column blocks
    (32 columns at VY = 2, 64 at VY = 4) holds a marked chunk in any of the strip's rows"""
    c, mark = marks_for(cfg_id, B=B)
    D = c["D"]
    SH = {64: 64, 128: 64, 256: 32}[D]
    CW = 32 if D <= 128 else 64
    NB = mark.shape[0]
    m = mark.reshape(NB, D, D // SH, SH, D // 32)              # [B, Dz, strips, rows, chunks]
    blk = m.any(3)                                             # [B, Dz, strips, chunks]
    if CW == 64:
        blk = blk.reshape(NB, D, D // SH, D // 64, 2).any(-1)
    live = blk.any(-1)
    print("cfg%d: live tiles %.3f of all; marked column blocks among live tiles %.3f; rows of a live tile with any mark %.3f" % (
        cfg_id, live.mean(), blk[live].mean(), m.any(-1)[live].mean()))


if __name__ == "__main__" and os.environ.get("YBLUR"):
    yblur_blocks(2)
    yblur_blocks(5, B=4)


def balance_static_radial(cfg_id, B=None):
    """static deal by distance from the image centre (snake over the SIMD groups of a view) against the Latin squares and a deal by measured cost"""
    c, mark = marks_for(cfg_id, B=B)
    D, K = c["D"], c["K"]; Dz = D; h = K // 2; G = K
    NB = mark.shape[0]
    NC, RG = D // 32, D // 4
    tile_any = mark.reshape(NB, Dz, RG, 4, NC).any(3)
    for name, T, look in (("zfwd", Dz + h, K), ("zbwd", Dz + 2 * h, 2 * K - 1)):
        ng = (T + G - 1) // G
        cost = np.zeros((NB, RG, NC))
        for g in range(ng):
            lo, hi = max(g * G - look, 0), min(g * G + G - 1, Dz - 1)
            dead = ~tile_any[:, lo:hi + 1].any(1) if hi >= lo else np.ones((NB, RG, NC), bool)
            cost += np.where(dead, 0.13, 1.0)
        cost /= ng
        flat = cost.reshape(NB, -1)                      # tile t = rg * NC + c
        NT = RG * NC
        rg, cc = np.divmod(np.arange(NT), NC)
        rad = ((rg + 0.5) * 4 - D / 2) ** 2 + ((cc + 0.5) * 32 - D / 2) ** 2
        def snake(order_per_view):
            gs = NT // 4
            sums = np.zeros((NB, gs))
            for b in range(NB):
                o = order_per_view[b]
                for r, t in enumerate(o):
                    q, pos = divmod(r, gs)
                    k = gs - 1 - pos if q & 1 else pos
                    sums[b, k] += flat[b, t]
            return sums / 4
        s_rad = snake([np.argsort(rad, kind="stable")] * NB)
        s_dyn = snake([np.argsort(-flat[b], kind="stable") for b in range(NB)])
        # global dynamic deal (all views together)
        allc = flat.reshape(-1); M = allc.size; gs = M // 4
        o = np.argsort(-allc, kind="stable"); sums = np.zeros(gs)
        for r, t in enumerate(o):
            q, pos = divmod(r, gs); k = gs - 1 - pos if q & 1 else pos; sums[k] += allc[t]
        print("cfg%d %s: mean %.3f | radial static: busiest SIMD %.3f | by measured cost per view %.3f | over all views %.3f" % (
            cfg_id, name, flat.mean(), s_rad.max(), s_dyn.max(), sums.max() / 4))


if __name__ == "__main__" and os.environ.get("RADIAL"):
    balance_static_radial(2)
    balance_static_radial(5, B=8)


def balance_within_wg(cfg_id, B=None):
    """the 16 tiles of a 1024-thread work-group (Latin squares) re-dealt INSIDE the work-group by measured cost (sort 16, snake over the
    4 SIMDs) -- no extra kernel, no global ranks: how close does that get to the global deal?"""
    c, mark = marks_for(cfg_id, B=B)
    D, K = c["D"], c["K"]; Dz = D; h = K // 2; G = K
    NB = mark.shape[0]
    NC, RG = D // 32, D // 4
    tile_any = mark.reshape(NB, Dz, RG, 4, NC).any(3)
    for name, T, look in (("zfwd", Dz + h, K), ("zbwd", Dz + 2 * h, 2 * K - 1)):
        ng = (T + G - 1) // G
        cost = np.zeros((NB, RG, NC))
        for g in range(ng):
            lo, hi = max(g * G - look, 0), min(g * G + G - 1, Dz - 1)
            dead = ~tile_any[:, lo:hi + 1].any(1) if hi >= lo else np.ones((NB, RG, NC), bool)
            cost += np.where(dead, 0.13, 1.0)
        cost /= ng
        mul2 = [0, 2, 3, 1]
        q = max(NC // 4, 1)
        nwg = (RG // 4) * q if NC >= 4 else RG // 8
        latin = np.zeros((NB, nwg, 4)); local = np.zeros((NB, nwg, 4))
        for g in range(nwg):
            tiles = []
            for w in range(16):
                s, j = w & 3, w >> 2
                a, b = s ^ j, mul2[s] ^ j
                if NC >= 4:
                    rg, cc = g // q + a * (RG // 4), g % q + q * b
                else:
                    rg, cc = g + a * (RG // 4) + (b >> 1) * (RG // 8), b & 1
                latin[:, g, s] += cost[:, rg, cc]
                tiles.append(cost[:, rg, cc])
            t = -np.sort(-np.stack(tiles, -1), axis=-1)          # [NB, 16] descending
            for r in range(16):
                qq, pos = divmod(r, 4)
                local[:, g, 3 - pos if qq & 1 else pos] += t[:, r]
        print("cfg%d %s: mean %.3f | Latin squares: busiest SIMD %.3f | re-dealt inside each work-group: %.3f | busiest work-group / 4: %.3f" % (
            cfg_id, name, cost.mean(), latin.max() / 4, local.max() / 4, latin.sum(-1).max() / 16))


if __name__ == "__main__" and os.environ.get("WITHIN"):
    balance_within_wg(2)
    balance_within_wg(5, B=8)
