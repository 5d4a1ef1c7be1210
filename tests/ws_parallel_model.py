"""TEST INFRASTRUCTURE: CPU model (numpy) of the parallel watershed formulation used by the HIP kernels.

Not the oracle and not the product: a readable model of the chain-key fixpoint
("Watershed") used to validate the theory against the oracle on small cases.
"""
import numpy as np

INF = np.uint64(0xFFFFFFFFFFFFFFFF)


def ordkey(v):
    v = np.asarray(v, np.float32) + np.float32(0.0)      # -0.0 -> +0.0
    u = v.view(np.uint32).astype(np.uint64)
    neg = (u >> np.uint64(31)) != 0
    return np.where(neg, (~u) & np.uint64(0xFFFFFFFF), u | np.uint64(0x80000000))


def model(p, depth=3, max_sweeps=100000):
    """p: dict from oracle.ws_oracle.prepare(). Returns padded flat labels."""
    v = ordkey(p["field"].ravel())
    mask = p["mask"] != 0
    out = p["out"].ravel().copy()
    N = v.size
    is_marker = out != 0
    active = mask & ~is_marker
    nbr, floc, bloc = p["nbr"], p["fwd_loc"].astype(np.int64), p["bwd_loc"].astype(np.int64)
    foff, boff = p["fwd_off"].astype(np.int64), p["bwd_off"].astype(np.int64)
    idx = np.arange(N, dtype=np.int64)

    def edges(src):
        """all (p, n) directed edges from source pixels `src` into active pixels"""
        P, Nn = [], []
        for i in range(nbr.size):
            n = src + nbr[i] + floc[i] * foff[src] + bloc[i] * boff[src]
            ok = active[n]
            P.append(src[ok]); Nn.append(n[ok])
        return np.concatenate(P), np.concatenate(Nn)

    src_all = idx[is_marker | active]
    EP, EN = edges(src_all)          # static edge list (p -> n), n active

    K2 = np.full(N, INF, np.uint64)
    K2[is_marker] = v[is_marker] << np.uint64(32)
    M1 = np.full(N, INF, np.uint64)
    sweeps = 0
    while True:
        kp = K2[EP]
        fin = kp != INF
        lp = kp >> np.uint64(32)
        gp = kp & np.uint64(0xFFFFFFFF)
        vn = v[EN]
        cand = np.where(vn > lp, (vn << np.uint64(32)) | np.uint64(1),
                        np.where(vn == lp, kp + np.uint64(1), kp))
        cand = np.where(fin, cand, INF)
        newK = K2.copy()
        np.minimum.at(newK, EN, cand)
        np.minimum.at(M1, EN, np.where(fin, kp, INF))
        sweeps += 1
        if np.array_equal(newK, K2):
            break
        K2 = newK
        assert sweeps < max_sweeps
    reached = active & (K2 != INF)
    entry = reached & ((K2 >> np.uint64(32)) == v) & ((K2 & np.uint64(0xFFFFFFFF)) == 1)
    # chain levels
    C = [K2]
    cand_edge = (K2[EP] == M1[EN]) & (K2[EP] != INF)
    EPc, ENc = EP[cand_edge], EN[cand_edge]
    ent_e = entry[ENc]
    match = np.ones(EPc.size, bool)
    total_sweeps = sweeps
    for k in range(1, depth):
        Ck = np.full(N, INF, np.uint64)
        Ck[is_marker] = 0
        while True:
            # offered value at level k: entry edge -> C[k-1](p); copy edge -> C[k](p)
            offered = np.where(ent_e, C[k - 1][EPc], Ck[EPc])
            offered = np.where(match, offered, INF)
            new = Ck.copy()
            np.minimum.at(new, ENc, offered)
            total_sweeps += 1
            if np.array_equal(new, Ck):
                break
            Ck = new
        C.append(Ck)
        offered = np.where(ent_e, C[k - 1][EPc], Ck[EPc])
        match = match & (offered == Ck[ENc])
    R = np.full(N, INF, np.uint64)
    R[is_marker] = np.arange(int(is_marker.sum()), dtype=np.uint64)   # idealised: push order
    mloc = idx[is_marker]
    # label set [lo, hi] over the roots of ALL fully matching candidates (the exactness check of the library)
    BIG = np.int64(1) << 40
    lo = np.full(N, BIG, np.int64)
    hi = np.full(N, -BIG, np.int64)
    lo[is_marker] = out[is_marker]
    hi[is_marker] = out[is_marker]
    while True:
        offered = np.where(match, R[EPc], INF)
        new = R.copy()
        np.minimum.at(new, ENc, offered)
        nlo = lo.copy()
        np.minimum.at(nlo, ENc, np.where(match, lo[EPc], BIG))
        nhi = hi.copy()
        np.maximum.at(nhi, ENc, np.where(match, hi[EPc], -BIG))
        total_sweeps += 1
        if np.array_equal(new, R) and np.array_equal(nlo, lo) and np.array_equal(nhi, hi):
            break
        R, lo, hi = new, nlo, nhi
    lab = out.copy()
    ok = reached & (R != INF)
    lab[ok] = out[mloc[R[ok].astype(np.int64)]]
    # report: bit 0 label depends on a last-resort tie-break; bit 1 origin between complete chains (equal-valued
    # markers); bit 2 origin between chains cut off at `depth`
    joins = match & ((lo[EPc] != lo[ENc]) | (hi[EPc] != hi[ENc]))
    complete = ((C[depth - 1] & np.uint64(0xFFFFFFFF)) == 0) if depth >= 2 else np.zeros(N, bool)
    origin = np.zeros(N, bool)
    origin[ENc[joins]] = True
    report = (ok & (lo != hi)).astype(np.uint8) | ((origin & complete).astype(np.uint8) << 1) \
        | ((origin & ~complete).astype(np.uint8) << 2)
    return lab, dict(sweeps_A=sweeps, sweeps_total=total_sweeps, report=report)


def run(fwd, bwd, field, markers, mask=None, conn=1, depth=3):
    from oracle import ws_oracle
    p = ws_oracle.prepare(fwd, bwd, field, markers, mask, conn)
    lab, info = model(p, depth)
    pd = p["pad"]

    def crop(a):
        o = a.reshape(p["out"].shape)
        return o[pd[0]:o.shape[0] - pd[0], pd[1]:o.shape[1] - pd[1], pd[2]:o.shape[2] - pd[2]].copy()

    info["report"] = crop(info["report"])
    return crop(lab), info


def run_auto(fwd, bwd, field, markers, mask=None, conn=1, depth=3, max_depth=12):
    """What the library does on its own: deepen until no origin is left by the depth cut-off."""
    while True:
        lab, info = run(fwd, bwd, field, markers, mask, conn, depth)
        info["depth"] = depth
        if not (info["report"] & 4).any() or depth >= max_depth:
            return lab, info
        depth += 1
