"""Summarise the kernel traces scripts/r05_overlap_timeline.sh takes: per configuration the wall time of a steady-state
step, the RCCL kernels of one step (count, busy time), the share of their busy time during which at least one compute
kernel was running (= hidden under compute), and the time during which ONLY communication kernels ran (= exposed)."""
import csv
import glob
import sys


def load(d):
    f = glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True)
    if not f:
        return None
    rows = [{"name": r["Kernel_Name"], "s": int(r["Start_Timestamp"]), "e": int(r["End_Timestamp"])}
            for r in csv.DictReader(open(f[0]))]
    rows.sort(key=lambda r: r["s"])
    return rows


def steady(rows):
    ad = [r for r in rows if "adamw_multi_kernel" in r["name"]]
    ends, prev = [], None
    for r in ad:
        if prev is not None and r["s"] - prev["e"] > 5e7:
            ends.append(prev["e"])
        prev = r
    ends.append(prev["e"])
    if len(ends) > 3:
        ends = ends[1:]                  # (the step behind the first optimizer run still warms allocator pools)
    t0, t1, n = ends[0], ends[-1], len(ends) - 1
    return [r for r in rows if r["s"] >= t0 and r["e"] <= t1], (t1 - t0) / n / 1e6, n


def union(iv):
    iv = sorted(iv)
    out = []
    for s, e in iv:
        if out and s <= out[-1][1]:
            out[-1][1] = max(out[-1][1], e)
        else:
            out.append([s, e])
    return out


def overlap(a, b):          # total length of the intersection of two unions of intervals
    i = j = 0
    tot = 0
    while i < len(a) and j < len(b):
        s, e = max(a[i][0], b[j][0]), min(a[i][1], b[j][1])
        if s < e:
            tot += e - s
        if a[i][1] < b[j][1]:
            i += 1
        else:
            j += 1
    return tot


def main(out):
    # (a 1-rank communicator's reduce-scatter is RCCL's `oneRankReduce` kernel — the pre-multiplied AVG over one rank — on
    #  the communication stream; its in-place all-gather is a no-op)
    is_comm = lambda n: "nccl" in n.lower() or "rccl" in n.lower() or "oneRank" in n
    print("# Flat data-parallel engine on a 1-rank RCCL group, collectives forced: overlap timeline of one MI355X step\n")
    print("| configuration | wall ms/step | RCCL kernels/step | RCCL busy ms/step | of it beside compute | comm-only ms/step | "
          "GEMM ms/step |")
    print("|---|---|---|---|---|---|---|")
    base = None
    for cfg in ("plain", "persistent", "per_tile"):
        rows = load(f"{out}/{cfg}")
        if rows is None:
            print(f"| {cfg} | (no trace) |")
            continue
        sel, wall, n = steady(rows)
        comm = [r for r in sel if is_comm(r["name"])]
        comp = [r for r in sel if not is_comm(r["name"])]
        cu, pu = union([(r["s"], r["e"]) for r in comm]), union([(r["s"], r["e"]) for r in comp])
        busy = sum(e - s for s, e in cu)
        hid = overlap(cu, pu)
        gemm = sum(r["e"] - r["s"] for r in comp if "tn::gemm" in r["name"])
        if cfg == "plain":
            base = wall
        label = {"plain": "plain step (no engine)", "persistent": "engine, persistent GEMM workgroups",
                 "per_tile": "engine, one GEMM workgroup per tile"}[cfg]
        tax = "" if cfg == "plain" or base is None else f" ({(wall / base - 1) * 100:+.1f} %)"
        print(f"| {label} | {wall:.1f}{tax} | {len(comm) / n:.0f} | {busy / n / 1e6:.2f} | "
              f"{(hid / busy * 100 if busy else 0):.0f} % | {(busy - hid) / n / 1e6:.2f} | {gemm / n / 1e6:.1f} |")
        if comm:
            names = {}
            for r in comm:
                k = r["name"][:70]
                names.setdefault(k, [0, 0])
                names[k][0] += 1
                names[k][1] += r["e"] - r["s"]
            for k, (c, t) in sorted(names.items(), key=lambda kv: -kv[1][1])[:4]:
                print(f"|   `{k}` | | {c / n:.0f} | {t / n / 1e6:.2f} | | | |")


    # kernel names that only the engine runs (top by time), for the record
    try:
        a, _, na = steady(load(f"{out}/plain"))
        b, _, nb = steady(load(f"{out}/per_tile"))
        ta, tb = {}, {}
        for rows, acc, n in ((a, ta, na), (b, tb, nb)):
            for r in rows:
                k = r["name"][:90]
                acc.setdefault(k, [0, 0.0])
                acc[k][0] += 1 / n
                acc[k][1] += (r["e"] - r["s"]) / n / 1e6
        print("\nkernels whose time per step differs most between the plain step and the engine (per-tile):\n")
        for k in sorted(set(ta) | set(tb), key=lambda k: -abs(tb.get(k, [0, 0])[1] - ta.get(k, [0, 0])[1]))[:14]:
            print(f"  {tb.get(k, [0, 0])[1] - ta.get(k, [0, 0])[1]:+8.2f} ms  plain {ta.get(k, [0, 0])[1]:8.2f} ({ta.get(k, [0, 0])[0]:5.0f}x)"
                  f"  engine {tb.get(k, [0, 0])[1]:8.2f} ({tb.get(k, [0, 0])[0]:5.0f}x)  {k}")
    except Exception as e:
        print("(no per-kernel comparison:", e, ")")


if __name__ == "__main__":
    main(sys.argv[1])
