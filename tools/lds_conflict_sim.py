"""LDS bank-conflict estimator for candidate padded layouts (design aid; MI355X LDS rules from
/opt/skills/guides/MI355X_MICROARCH.md §LDS): ds_read_b64 = 2 groups of 32 lanes, bank pair =
(word8 mod 32); ds_write_b64 = 4 groups of 16 lanes, word8 mod 16."""
import itertools, sys
sys.path.insert(0, '/root/repo/tools')
from ntt_layout_proto import brv


def cost(addrs, kind):
    """addrs: 64 word8 addresses. returns LDS cycles relative to conflict-free (1.0)."""
    if kind == 'r':
        groups = [addrs[0:32], addrs[32:64]]; mod = 32
    else:
        groups = [addrs[i:i + 16] for i in range(0, 64, 16)]; mod = 16
    tot = 0
    for g in groups:
        banks = {}
        for a in set(g):
            banks.setdefault(a % mod, set()).add(a)
        tot += max(len(v) for v in banks.values())
    return tot / len(groups)


def evaluate(phi, n=14, T=1024, passes=(4, 4, 4, 2), bitrev_last=True):
    e = n - (T.bit_length() - 1)
    res = {}
    s0 = 0
    for pi, k in enumerate(passes):
        sets = 1 << (e - k)
        lo_bits = n - s0 - k
        last = pi == len(passes) - 1
        worst_r = worst_w = 0
        for wave in (0, 5, T // 64 - 1):
            for u in range(sets):
                for r in range(1 << k):
                    addrs = []
                    for lane in range(64):
                        tid = wave * 64 + lane
                        c = u * T + tid
                        if last and bitrev_last:
                            c = brv(c, n - k)
                        lo = c & ((1 << lo_bits) - 1); hi = c >> lo_bits
                        addrs.append(phi((hi << (n - s0)) + (r << lo_bits) + lo))
                    worst_r = max(worst_r, cost(addrs, 'r')); worst_w = max(worst_w, cost(addrs, 'w'))
        res[pi] = (worst_r, worst_w)
        s0 += k
    return res


if __name__ == '__main__':
    cands = {}
    for sa, pa, sb, pb in itertools.product((4, 5, 6), (0, 1, 2, 4), (8, 9, 10), (0, 1, 2)):
        phi = lambda j, sa=sa, pa=pa, sb=sb, pb=pb: j + pa * (j >> sa) + pb * (j >> sb)
        r = evaluate(phi)
        # pass0: write only; last: read only; others both
        score = r[0][1] + r[1][0] + r[1][1] + r[2][0] + r[2][1] + r[3][0]
        size = phi((1 << 14) - 1) + 1
        cands[(sa, pa, sb, pb)] = (score, size, r)
    for k, v in sorted(cands.items(), key=lambda kv: kv[1][0])[:12]:
        print(k, 'score', v[0], 'words', v[1], v[2])
