"""Per-kernel difference of two optimizer steps taken from rocprofv3 kernel traces (`--kernel-trace --output-format csv`): a step = the
dispatches between two AdamW launches.  python tools/trace_diff.py A_kernel_trace.csv stepA B_kernel_trace.csv stepB [min_us]
(profiles/r03_cls_only_trace_diff.txt: the every-row step against the [CLS]-rows step of ONE bench.py process -- its legs run back to back;
profiles/r03_forced_ddp_trace_diff.txt: CLIMB_AMD_FORCE_DDP=1 against the plain step)."""
import collections
import csv
import sys


def step_of(path, k):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(rows) if "adamw" in r["Kernel_Name"]]
    step = rows[idx[k] + 1:idx[k + 1] + 1]
    d, n = collections.Counter(), collections.Counter()
    for r in step:
        name = r["Kernel_Name"][:84]
        d[name] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        n[name] += 1
    wall = (int(step[-1]["End_Timestamp"]) - int(step[0]["Start_Timestamp"])) / 1e3
    return d, n, wall, len(step)


def main():
    a, ka, b, kb = sys.argv[1], int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
    floor = float(sys.argv[5]) if len(sys.argv) > 5 else 3.0
    da, na, wa, la = step_of(a, ka)
    db, nb, wb, lb = step_of(b, kb)
    print(f"A = {a} step {ka}: {wa:9.1f} us wall, {la} launches, kernels sum to {sum(da.values()):9.1f} us")
    print(f"B = {b} step {kb}: {wb:9.1f} us wall, {lb} launches, kernels sum to {sum(db.values()):9.1f} us")
    print(f"{'A - B us':>9s} {'A us':>9s} {'(n)':>5s} {'B us':>9s} {'(n)':>5s}  kernel")
    for k in sorted(set(da) | set(db), key=lambda k: -(da[k] - db[k])):
        if abs(da[k] - db[k]) >= floor:
            print(f"{da[k] - db[k]:9.1f} {da[k]:9.1f} {na[k]:5d} {db[k]:9.1f} {nb[k]:5d}  {k}")


if __name__ == "__main__":
    main()
