"""Order of the factor's columns in LDS for the row kernel (lmpc_row_kernel.hpp, rowp_cbm): a backtracking search for an
order in which the 16 columns of each slot of positions start at 16 different offsets modulo 16, so that column accesses
(lane i reads its own column) are free of LDS bank conflicts.  python tools/row_layout.py [rows]"""
import random, sys, time
sys.setrecursionlimit(10000)


def solve(capp, seed=0, tlimit=30):
    ncol = capp - 1
    p0 = lambda t: t & ~3
    lrow = capp if capp % 2 else capp + 1                # (even row counts: columns padded by one entry, odd lengths)
    L = [lrow - p0(t) for t in range(ncol)]
    rnd = random.Random(seed)
    used = [set() for _ in range((ncol + 15) // 16)]
    c, remaining, t0 = {}, set(range(ncol)), time.time()

    def dfs(acc):
        if time.time() - t0 > tlimit:
            return False
        if not remaining:
            return True
        cands = list(remaining)
        rnd.shuffle(cands)
        tried = set()
        for t in cands:
            key = (L[t], p0(t), t >> 4)
            if key in tried:
                continue
            tried.add(key)
            r, g = (acc - p0(t)) % 16, t >> 4
            if r in used[g]:
                continue
            used[g].add(r); remaining.discard(t); c[t] = acc
            if dfs(acc + L[t]):
                return True
            used[g].discard(r); remaining.add(t); del c[t]
        return False

    return dict(c) if dfs(0) else None


if __name__ == "__main__":
    capp = int(sys.argv[1]) if len(sys.argv) > 1 else 31
    for seed in range(5):
        c = solve(capp, seed)
        if c:
            print("rows", capp, "size", max(c[t] + (capp if capp % 2 else capp + 1) - (t & ~3) for t in c), "cbm =", [c[t] - (t & ~3) for t in range(capp - 1)])
            break
    else:
        print("rows", capp, ": no conflict-free order (all column lengths a multiple of four?)")
