"""Condenses rocprofv3's per-kernel statistics (--kernel-trace --stats --output-format csv) into the table committed under
profiles/: one row per kernel with ms per step, share, launches per step and the average launch duration.

  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d <dir> -o s -- python3 /root/repo/bench.py --steps 20 --warmup 5 --no-infer
  python tools/kernel_stats.py <dir> <steps incl. warm-up> "<header line>" > profiles/rNN_kernel_stats.csv
"""
import csv, glob, os, re, sys


def main():
    d, steps = sys.argv[1], float(sys.argv[2])
    header = sys.argv[3] if len(sys.argv) > 3 else ""
    paths = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)
    if not paths:
        raise SystemExit("no *kernel_stats.csv under " + d)
    rows = []
    for path in paths:
        with open(path, newline="") as f:
            for r in csv.DictReader(f):
                rows.append((r["Name"], int(r["Calls"]), float(r["TotalDurationNs"]), float(r["AverageNs"])))
    tot = sum(r[2] for r in rows)
    if header:
        print("# " + header)
    print(f"# totals divided by {steps:g} steps (warm-up included); kernels of concurrent streams overlap, so the sum "
          f"({tot / steps / 1e6:.2f} ms) may exceed the step time")
    print("ms_per_step,percent,calls_per_step,avg_us,kernel")
    for name, calls, total, avg in sorted(rows, key=lambda r: -r[2]):
        name = re.sub(r"\s+", " ", name)
        print(f"{total / steps / 1e6:.3f},{100 * total / tot:.2f},{calls / steps:.1f},{avg / 1e3:.1f},\"{name}\"")


if __name__ == "__main__":
    main()
