"""From a rocprofv3 kernel trace: the adamw launches of the LAST step (start, duration, queue), and for the window they
cover the GEMM kernels' durations against the same kernels' durations in a window without them."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
K = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")) for r in rows]
K.sort()
ad = [k for k in K if "adamw_multi" in k[2]]
print("adamw launches:", len(ad))
# steps = clusters of adamw launches separated by > 100 ms
steps, cur = [], [ad[0]]
for k in ad[1:]:
    if k[0] - cur[-1][1] > 100e6:
        steps.append(cur); cur = [k]
    else:
        cur.append(k)
steps.append(cur)
for s in steps[-2:]:
    t0, t1 = s[0][0], max(k[1] for k in s)
    busy = sum(k[1] - k[0] for k in s)
    print(f"step: {len(s)} adamw launches, window {1e-6*(t1-t0):.2f} ms, sum of durations {1e-6*busy:.2f} ms, queues {sorted(set(k[3] for k in s))}")
    inside = [k for k in K if "adamw" not in k[2] and k[0] < t1 and k[1] > t0]
    byname = collections.Counter()
    for k in inside:
        byname[k[2][:60]] += 1e-6 * (min(k[1], t1) - max(k[0], t0))
    print("   other kernels overlapping the window:", len(inside), "covering", round(sum(byname.values()), 2), "ms")
    for n, v in byname.most_common(6):
        print(f"      {v:8.2f} ms  {n}")
# forward GEMM durations inside vs outside the adamw windows (same kernel name)
wins = [(s[0][0], max(k[1] for k in s)) for s in steps]
def inwin(k):
    return any(k[0] < b and k[1] > a for a, b in wins)
stat = collections.defaultdict(lambda: [[], []])
for k in K:
    if "gemm_kernel<false, false" in k[2]:
        stat[k[2][:70]][0 if inwin(k) else 1].append(1e-3 * (k[1] - k[0]))
for n, (a, b) in stat.items():
    if a and b:
        print(f"{n}: inside adamw windows n={len(a)} mean {sum(a)/len(a):.1f} us | outside n={len(b)} mean {sum(b)/len(b):.1f} us")
