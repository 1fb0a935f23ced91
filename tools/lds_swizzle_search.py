"""Enumerates LDS layouts of the fp32 halo-patch staging area of k_qkv2 (csrc/ira_qkv2.hip) that are a pure function of the linear
16-byte piece index n (so that the C waves' write addresses need no division) and counts the bank conflicts of the P waves' fragment-shaped
ds_read_b128 (16-lane service groups of MI355X_MICROARCH.md, LDS table).  Result used by the kernel: n ^ ((n >> 3) & 15) at C = 96 and
n ^ ((n >> 2) & 15) at C = 48 are as good as one pad piece per pixel (1.55 LDS cycles per group on average; the halo-column group is the
outlier) and need no extra space."""
PW = 34; PH = 10; TH = 8


def rowperm(i): return 16 * (i >> 4) + 8 * ((i >> 2) & 1) + 4 * ((i >> 3) & 1) + (i & 3)


def group_px(g, p):
    if g < TH: return (g + 1) * PW + 1 + p
    if g == TH: return 1 + p
    if g == TH + 1: return (PH - 1) * PW + 1 + p
    if p < PH: return p * PW
    if p < 2 * PH: return (p - PH) * PW + PW - 1
    return -1


GROUPS128 = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]


def cost(addr16, C):
    PP = C // 4
    tot = cnt = worst = 0
    for g in range(11):
        for s in range(C // 16):
            for hk in range(2):
                for half in range(2):
                    for grp in GROUPS128:
                        banks = {}
                        for r in grp:
                            q = max(group_px(g, rowperm(r)), 0)
                            b = addr16(q * PP + 4 * s + 2 * hk + half) % 16
                            banks[b] = banks.get(b, 0) + 1
                        m = max(banks.values()); tot += m; cnt += 1; worst = max(worst, m)
    return round(tot / cnt, 3), worst


if __name__ == '__main__':
    for C in (96, 48):
        PP = C // 4
        print('C =', C, ' one pad piece per pixel:', cost(lambda n: n + n // PP, C), ' no padding:', cost(lambda n: n, C))
        for a in range(1, 8):
            print('   n ^ ((n >> %d) & 15):' % a, cost(lambda n: n ^ ((n >> a) & 15), C), '   n + (n >> %d):' % a, cost(lambda n: n + (n >> a), C))
