import sys, numpy as np
sys.path.insert(0, "/root/repo")
from __graft_entry__ import load_package
from oracle import cref as O
eng = load_package().Engine(0)
n, t, S, G = 31, 10, 21, 97
ids = [int(i) for i in np.random.default_rng(S).permutation(n)[:S]]
for true_deg in (10, 20, 0):
    co = O.fill_random(300 + true_deg, G * (true_deg + 1)).reshape(G, true_deg + 1, 4)
    co[3] = 0; co[4, true_deg] = 0
    rc, sh = O.compute_shares(co, n, true_deg)
    ev = np.ascontiguousarray(sh[ids])
    rc, got, deg = eng.batch_interpolate(ids, ev, n)
    bad = [g for g in range(G) if not np.array_equal(got[g, :true_deg + 1], co[g])]
    print(true_deg, rc, "bad chunks:", bad[:10], len(bad))
    for g in bad[:2]:
        print(" got", O.u256_to_ints(got[g])[:3], "\n want", O.u256_to_ints(co[g])[:3])
        rc0, want, sec = O.nonrobust_recover_secret(ids, [S - 1] * S, ev[:, g], n)
        print(" oracle", O.u256_to_ints(want)[:3] if len(want) else [])
