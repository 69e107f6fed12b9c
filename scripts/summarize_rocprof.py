"""Turn a rocprofv3 `--kernel-trace --output-format csv` trace of bench.py into the per-step, per-category
markdown summary kept under profiles/ (steady state = between the ends of the first and last optimizer runs)."""
import csv
import sys
from collections import defaultdict


def category(name):
    if name.startswith("Cijk") or name.startswith("Custom_Cijk"):
        return "GEMM (hipBLASLt, via F.linear)"
    if "tn::gemm::" in name:
        return "tn::gemm hand-written MFMA GEMM (HIP)"
    if "attn_" in name:
        return "tn:: packed attention (HIP)"
    if "adamw" in name or "sumsq" in name:
        return "tn:: clip + AdamW (HIP)"
    if "tn::" in name:
        return "tn:: other hand-written HIP kernels"
    if any(k in name for k in ("conv", "igemm", "Im2d2Col", "transpose", "SubTensor")):
        return "MIOpen conv stem"
    if "at::native" in name or "rocprim" in name:
        return "torch aten glue"
    return "other"


def main(path, out):
    rows = list(csv.DictReader(open(path)))
    for r in rows:
        r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    adam = sorted((r for r in rows if "adamw_kernel" in r["Kernel_Name"] or "adamw_multi_kernel" in r["Kernel_Name"]), key=lambda r: r["s"])
    # step boundaries = gaps > 50 ms between adamw launches
    ends, prev = [], None
    for r in adam:
        if prev is not None and r["s"] - prev["e"] > 5e7:
            ends.append(prev["e"])
        prev = r
    ends.append(prev["e"])
    t0, t1, nsteps = ends[0], ends[-1], len(ends) - 1
    sel = [r for r in rows if r["s"] >= t0 and r["e"] <= t1]
    per_k, per_c = defaultdict(lambda: [0, 0.0]), defaultdict(float)
    for r in sel:
        d = (r["e"] - r["s"]) / 1e6
        k = r["Kernel_Name"]
        per_k[k][0] += 1
        per_k[k][1] += d
        per_c[category(k)] += d
    tot = sum(per_c.values())
    with open(out, "w") as f:
        f.write(f"# rocprofv3 kernel trace — steady state, {nsteps} step(s), wall {(t1 - t0) / 1e6 / nsteps:.1f} ms/step "
                f"(profiled), kernel time {tot / nsteps:.1f} ms/step\n\n")
        f.write("| category | ms/step | share |\n|---|---|---|\n")
        for c, v in sorted(per_c.items(), key=lambda x: -x[1]):
            f.write(f"| {c} | {v / nsteps:.2f} | {100 * v / tot:.1f} % |\n")
        f.write("\n| kernel | launches/step | avg µs | ms/step |\n|---|---|---|---|\n")
        for k, (n, v) in sorted(per_k.items(), key=lambda x: -x[1][1])[:45]:
            f.write(f"| `{k[:110]}` | {n / nsteps:.1f} | {v / n * 1e3:.1f} | {v / nsteps:.2f} |\n")
    print(open(out).read()[:3000])


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
