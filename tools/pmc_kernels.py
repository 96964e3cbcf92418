"""Per-kernel averages of arbitrary rocprofv3 counters, one directory per counter pass:

  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY ... --output-format csv -d <dir1> -o p -- python3 tools/one_conv.py ...
  python tools/pmc_kernels.py <dir1> [<dir2> ...] [--match substr]

Prints, per kernel name, the number of dispatches and every counter's mean per dispatch (summed over the XCDs /
dimensions rocprofv3 reports), plus the mean dispatch duration of that pass."""
import csv, glob, os, re, sys
from collections import defaultdict


def main():
    dirs = [a for a in sys.argv[1:] if not a.startswith("--")]
    match = None
    if "--match" in sys.argv:
        match = sys.argv[sys.argv.index("--match") + 1]
        dirs = [d for d in dirs if d != match]
    agg = defaultdict(lambda: defaultdict(lambda: [0.0, set()]))
    dur = defaultdict(lambda: [0.0, set()])
    for d in dirs:
        for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(path, newline="") as f:
                for row in csv.DictReader(f):
                    name = re.sub(r"^void ", "", row["Kernel_Name"].replace("(anonymous namespace)::", ""))
                    if match and match not in name:
                        continue
                    key = (path, row["Dispatch_Id"])
                    a = agg[name][row["Counter_Name"]]
                    a[0] += float(row["Counter_Value"])
                    a[1].add(key)
                    if key not in dur[name][1] and row.get("End_Timestamp"):
                        dur[name][0] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
                        dur[name][1].add(key)
    for name, cs in sorted(agg.items(), key=lambda kv: -dur[kv[0]][0]):
        n = max(1, len(dur[name][1]))
        print(f"{name[:110]}  dispatches {n}  avg {dur[name][0] / n / 1e3:.1f} us (under the profiler)")
        for c, (v, keys) in sorted(cs.items()):
            print(f"    {c:32s} {v / max(1, len(keys)):.4e}")


if __name__ == "__main__":
    main()
