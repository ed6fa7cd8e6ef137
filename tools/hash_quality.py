import numpy as np
M=np.uint64(0xFFFFFFFF)
def lowbias32(x):
    x = x.astype(np.uint64) & M
    x ^= x >> np.uint64(16); x = (x * np.uint64(0x7feb352d)) & M
    x ^= x >> np.uint64(15); x = (x * np.uint64(0x846ca68b)) & M
    x ^= x >> np.uint64(16)
    return x
def key(seed):
    s = np.uint64(seed)
    k = ((s & M) * np.uint64(0x9E3779B1) & M) ^ (s >> np.uint64(32))
    return lowbias32(np.array([k]))[0]
def h(seed, idx):
    idx = idx.astype(np.uint64)
    lo, hi = idx & M, idx >> np.uint64(32)
    rot = ((hi << np.uint64(16)) | (hi >> np.uint64(16))) & M
    return lowbias32(((lo ^ rot) + key(seed)) & M)
n = 1<<22
idx = np.arange(n)
for p in (0.1,0.2,0.5):
  for seed in (1234, 1235, 77, 2**40+5):
    u = (h(seed, idx) >> np.uint64(8)).astype(np.float64) / 16777216.0
    keep = u >= p
    sig = np.sqrt(p*(1-p)/n)
    # correlation with neighbouring seed and neighbouring index
    u2 = (h(seed+1, idx) >> np.uint64(8)).astype(np.float64) / 16777216.0
    k2 = u2 >= p
    c_seed = np.corrcoef(keep, k2)[0,1]; c_idx = np.corrcoef(keep[:-1], keep[1:])[0,1]; c_row=np.corrcoef(keep[:-128], keep[128:])[0,1]
    print(f"p={p} seed={seed}: rate err {(keep.mean()-(1-p))/sig:+.2f} sigma  corr(seed+1) {c_seed:+.4f} corr(idx+1) {c_idx:+.4f} corr(idx+128) {c_row:+.4f}")
# box-muller
u1 = ((h(77, 2*idx)>>np.uint64(8)).astype(np.float64)+0.5)/16777216.0; u2=((h(77,2*idx+1)>>np.uint64(8)).astype(np.float64)+0.5)/16777216.0
z = np.sqrt(-2*np.log(u1))*np.cos(2*np.pi*u2)
print("randn mean %.4f std %.4f m4 %.4f skew %.4f" % (z.mean(), z.std(), (z**4).mean(), (z**3).mean()))
# chi2 of 256 buckets on top byte
b = (h(5, idx) >> np.uint64(24)).astype(int); c = np.bincount(b, minlength=256); print("chi2/255 =", ((c-n/256)**2/(n/256)).sum()/255)
# bit balance
x = h(9, idx); print("bit means", [round(float(((x>>np.uint64(k))&np.uint64(1)).mean()),4) for k in range(8,32,3)])
def triple32(x):
    x = x.astype(np.uint64) & M
    x ^= x >> np.uint64(17); x = (x * np.uint64(0xed5ad4bb)) & M
    x ^= x >> np.uint64(11); x = (x * np.uint64(0xac4c1b51)) & M
    x ^= x >> np.uint64(15); x = (x * np.uint64(0x31848bab)) & M
    x ^= x >> np.uint64(14)
    return x
rng = np.random.default_rng(0)
for name, f in (("lowbias32", lowbias32), ("triple32", triple32), ("weyl+lowbias32", lambda x: lowbias32((x.astype(np.uint64)*np.uint64(0x9E3779B1))&M)), ("numpy random", None)):
    res=[]
    for off in (0, 12345678, 2**31-5, 777):
        if f is None: x = rng.integers(0, 2**32, n, dtype=np.uint64)
        else: x = f((idx.astype(np.uint64) + np.uint64(off)) & M)
        r=[]
        for sh in (24, 8, 0):
            b = ((x >> np.uint64(sh)) & np.uint64(255)).astype(int); c = np.bincount(b, minlength=256); r.append(((c-n/256)**2/(n/256)).sum()/255)
        res.append(np.round(r,2))
    print(name, res)

# ---- round 6: the shipped form keys the counter AND a middle round (common.h: mix32k).  Window-overlap check: two
# seeds whose first keys lie d apart gave, with the additive key alone, masks that are copies of each other shifted by d.
def key2(seed):
    s = np.uint64(seed)
    k = triple32(np.array([((s & M) * np.uint64(0x9E3779B1) & M) ^ (s >> np.uint64(32))]))[0]
    return k, triple32(np.array([k ^ np.uint64(0x85ebca6b)]))[0]
def mix32k(x, k2):
    x = x.astype(np.uint64) & M
    x ^= x >> np.uint64(17); x = (x * np.uint64(0xed5ad4bb)) & M
    x ^= x >> np.uint64(11); x = (x + k2) & M; x = (x * np.uint64(0xac4c1b51)) & M
    x ^= x >> np.uint64(15); x = (x * np.uint64(0x31848bab)) & M
    x ^= x >> np.uint64(14)
    return x
def h6(seed, idx, keyed=True):
    k, k2 = key2(seed)
    x = (idx.astype(np.uint64) + k) & M
    return mix32k(x, k2 if keyed else np.uint64(0))
ks = {s: int(key2(s)[0]) for s in range(1, 40000)}
order = sorted(ks, key=ks.get)
best = min(zip(order[:-1], order[1:]), key=lambda ab: ks[ab[1]] - ks[ab[0]])
d = ks[best[1]] - ks[best[0]]
print(f"closest first keys among seeds 1..39999: seeds {best} are {d} counters apart")
for keyed in (False, True):
    a = (h6(best[0], idx + np.uint64(d), keyed) >> np.uint64(8)).astype(np.float64) / 16777216.0 >= 0.1
    b = (h6(best[1], idx, keyed) >> np.uint64(8)).astype(np.float64) / 16777216.0 >= 0.1
    print(f"  middle round keyed = {keyed}: corr(mask(seed {best[0]})[i + d], mask(seed {best[1]})[i]) = {np.corrcoef(a, b)[0, 1]:+.4f}")
for p in (0.1, 0.5):
    for seed in (1234, 1235, 2**40 + 5):
        u = (h6(seed, idx) >> np.uint64(8)).astype(np.float64) / 16777216.0
        keep = u >= p
        sig = np.sqrt(p * (1 - p) / n)
        k2_ = (h6(seed + 1, idx) >> np.uint64(8)).astype(np.float64) / 16777216.0 >= p
        print(f"shipped form p={p} seed={seed}: rate err {(keep.mean() - (1 - p)) / sig:+.2f} sigma  corr(seed+1) {np.corrcoef(keep, k2_)[0, 1]:+.4f} "
              f"corr(idx+1) {np.corrcoef(keep[:-1], keep[1:])[0, 1]:+.4f} corr(idx+128) {np.corrcoef(keep[:-128], keep[128:])[0, 1]:+.4f}")
b = (h6(5, idx) >> np.uint64(24)).astype(int); c = np.bincount(b, minlength=256); print("shipped form chi2/255 (top byte) =", ((c - n / 256) ** 2 / (n / 256)).sum() / 255)
u1 = ((h6(77, 2 * idx) >> np.uint64(8)).astype(np.float64) + 0.5) / 16777216.0; u2 = ((h6(77, 2 * idx + 1) >> np.uint64(8)).astype(np.float64) + 0.5) / 16777216.0
z = np.sqrt(-2 * np.log(u1)) * np.cos(2 * np.pi * u2)
print("shipped form randn mean %.4f std %.4f m4 %.4f skew %.4f" % (z.mean(), z.std(), (z ** 4).mean(), (z ** 3).mean()))
