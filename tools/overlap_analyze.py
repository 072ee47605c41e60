"""rocprofv3 (rocpd sqlite) kernel + memory-copy trace of tools/overlap_run.py -> how much of the exchange copies' time
ran beside compute kernels.  Prints the schema objects it used (the rocpd views differ between ROCm releases)."""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
objs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]


def cols(t):
    return [r[1] for r in c.execute("pragma table_info('%s')" % t)]


def pick(cands):
    for t in cands:
        if t in objs:
            return t
    return None


kt = pick(["kernels", "rocpd_kernel_dispatch"]) or next((o for o in objs if "kernel" in o.lower() and "top" not in o.lower()), None)
mt = pick(["memory_copies", "rocpd_memory_copy"]) or next((o for o in objs if "memory_cop" in o.lower() or "memcpy" in o.lower()), None)
print("objects:", ", ".join(o for o in objs if not o.startswith("sqlite"))[:1500])
print("\nkernel trace object:", kt, cols(kt) if kt else None)
print("memory copy object:", mt, cols(mt) if mt else None)
if not kt or not mt:
    sys.exit(0)
kc, mc = cols(kt), cols(mt)
ks, ke = ("start", "end") if "start" in kc else ("start_timestamp", "end_timestamp")
ms, me = ("start", "end") if "start" in mc else ("start_timestamp", "end_timestamp")
kname = "name" if "name" in kc else ("kernel_name" if "kernel_name" in kc else kc[0])
ALL = list(c.execute("select %s, %s, %s from %s" % (ks, ke, kname, kt)))
K = [(s, e, n) for s, e, n in ALL if "fpm::" in str(n)]
# device-to-device hipMemcpyAsync runs as a copy KERNEL here (__amd_rocclr_copyBuffer), not on an SDMA engine: the exchange
# pieces are those lasting more than 20 us (the flag read-backs and small halo rows are shorter)
CK = [(s, e) for s, e, n in ALL if "copyBuffer" in str(n) and e - s > 20000]
M = list(c.execute("select %s, %s from %s" % (ms, me, mt)))
size_col = next((x for x in mc if x in ("size", "bytes", "size_bytes")), None)
if size_col:
    Msz = list(c.execute("select %s, %s, %s from %s" % (ms, me, size_col, mt)))
    M = [(s, e) for s, e, b in Msz if b and b >= (1 << 20)]          # the exchange pieces, not the flag read-backs
print("\n%d fpm kernels, %d large device copies" % (len(K), len(M)))
K.sort()
# union of kernel intervals
U = []
for s, e, _ in K:
    if U and s <= U[-1][1]:
        U[-1][1] = max(U[-1][1], e)
    else:
        U.append([s, e])
tot = ov = 0
for s, e in M:
    tot += e - s
    for a, b in U:
        if b <= s:
            continue
        if a >= e:
            break
        ov += min(b, e) - max(a, s)
span = (max(e for _, e, _ in K) - min(s for s, _, _ in K)) if K else 0
print("copy time %.3f ms, of which %.3f ms (%.1f %%) ran while an fpm kernel was executing; kernels busy %.3f ms of a %.3f ms span"
      % (tot / 1e6, ov / 1e6, 100.0 * ov / tot if tot else 0, sum(b - a for a, b in U) / 1e6, span / 1e6))

tot = ov = 0
for s, e in CK:
    tot += e - s
    for a, b in U:
        if b <= s:
            continue
        if a >= e:
            break
        ov += min(b, e) - max(a, s)
print("%d exchange copy kernels (> 20 us): %.3f ms, of which %.3f ms (%.1f %%) ran while an fpm kernel was executing"
      % (len(CK), tot / 1e6, ov / 1e6, 100.0 * ov / tot if tot else 0))
